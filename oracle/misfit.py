"""CPU restatement of the reference's data-misfit expressions -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (physicsbasedfwi2_amd/) never does.

Follows, line by line:
* models/networks.py:5418-5419  observed data: max over time of |d| per trace, d / (max + 1e-10)
* models/networks.py:5467-5472  predicted data minus the direct wave, same normalisation
* models/networks.py:5476       torch.nn.L1Loss() = mean of |a - b| over every element
* models/networks.py:5491       lossinner.backward(): the gradient reaching the propagator output
* seisgan/fwi/layers.py:176-178 objective 0.5 * ||pred - obs||^2, adjoint source pred - obs

Pinned by tests/test_misfit_oracle.py against torch autograd evaluating the very expressions of
networks.py:5467-5476 on seeded inputs (the surrounding prop() cannot be imported: deepwave).
"""
import numpy as np

EPS = 1e-10


def trace_normalize(d):
    m = np.abs(d).max(axis=0, keepdims=True)
    return d / (np.abs(m) + EPS)


def l1_trace_normalized(pred, obs_norm, direct=None, dtype=np.float64):
    """Returns (loss, dloss/dpred).  Arrays are [nt, ...]; time on axis 0."""
    pred = np.asarray(pred, dtype=dtype)
    obs = np.asarray(obs_norm, dtype=dtype)
    d = pred if direct is None else pred - np.asarray(direct, dtype=dtype)
    nt = d.shape[0]
    flat = d.reshape(nt, -1)
    o = obs.reshape(nt, -1)
    a = np.abs(flat)
    arg = a.argmax(axis=0)                      # first maximum, as torch.max(dim=0) on ties
    m = a[arg, np.arange(flat.shape[1])]
    inv = 1.0 / (m + EPS)
    r = flat * inv - o
    n = flat.size
    loss = np.abs(r).sum() / n
    g_dn = np.sign(r) / n                       # d mean|.| / d dn
    adj = g_dn * inv
    # the maximum feeds every sample of its trace: d dn_t / d m = -d_t / (m+eps)^2, dm/dd_t* = sign(d_t*)
    c = (g_dn * flat).sum(axis=0)
    cols = np.arange(flat.shape[1])
    adj[arg, cols] -= np.sign(flat[arg, cols]) * c * inv * inv
    return dtype(loss), adj.reshape(d.shape)


def l2_half(pred, obs, dtype=np.float64):
    r = np.asarray(pred, dtype=dtype) - np.asarray(obs, dtype=dtype)
    return dtype(0.5 * (r * r).sum()), r


def global_correlation(pred, obs, dtype=np.float64):
    """DENISE's global-correlation norm (Choi & Alkhalifah 2012): per trace (time on axis 0)
    -<s, o> / (|s| |o|), summed over traces; a dead trace on either side contributes nothing.
    Returns (loss, dloss/dpred)."""
    s = np.asarray(pred, dtype=dtype)
    o = np.asarray(obs, dtype=dtype)
    nt = s.shape[0]
    sf, of = s.reshape(nt, -1), o.reshape(nt, -1)
    ss, oo, so = (sf * sf).sum(0), (of * of).sum(0), (sf * of).sum(0)
    ok = (ss > 0) & (oo > 0)
    den = np.where(ok, np.sqrt(ss) * np.sqrt(oo), 1.0)
    c = np.where(ok, so / den, 0.0)
    adj = np.where(ok, -1.0 / den, 0.0) * of + np.where(ok, c / np.where(ok, ss, 1.0), 0.0) * sf
    return dtype(-c.sum()), adj.reshape(s.shape)
