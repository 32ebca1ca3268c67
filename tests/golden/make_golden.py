#!/usr/bin/env python3
"""Mint golden vectors from the reference's OWN numpy helper code (run in the build
container only; /root/reference never travels to the GPU box).

What is executed: the pure-numpy bodies of
  seisgan/fwi/pde/seismic/model.py   damp_boundary (6-29), Model.critical_dt (160-168),
                                     Model.pad (194-200), Model.vp setter (182-192)
  seisgan/fwi/pde/seismic/source.py  TimeAxis (18-69), RickerSource.wavelet (224-231)
Those files import devito / cached_property at module top, which are not installed, so
empty placeholder modules are registered first purely to let the `import` lines succeed
(SURVEY.md section 8c); no devito behaviour is emulated -- the functions called below never
touch a devito object (methods are invoked on a plain namespace standing in for `self`).

Also stored: input/output pairs of the plain torch expressions of the misfit /
gradient-conditioning blocks of models/networks.py (5418-5419, 5467-5476, 5492-5493,
7808-7862), re-typed here because `models.networks` cannot be imported (deepwave import at
line 10).

Output: tests/golden/seisgan_helpers.npz, tests/golden/prop_expressions.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    class _Nothing(object):
        pass

    _placeholder("devito", Grid=_Nothing, Function=_Nothing, Constant=_Nothing,
                 Dimension=_Nothing)
    _placeholder("devito.function", SparseTimeFunction=_Nothing)
    _placeholder("devito.logger", error=print)
    _placeholder("cached_property", cached_property=property)

    model = _load(os.path.join(REF, "seisgan/fwi/pde/seismic/model.py"), "ref_model")
    source = _load(os.path.join(REF, "seisgan/fwi/pde/seismic/source.py"), "ref_source")

    out = {}
    # damping fields -----------------------------------------------------------------------
    for tag, shape, nb, sp in [("a", (240, 240), 20, (10.0, 10.0)),
                               ("b", (70, 90), 10, (15.0, 15.0)),
                               ("c", (64, 48), 5, (10.0, 20.0)),
                               ("d", (130, 130), 40, (0.5, 0.5))]:
        damp = np.zeros(shape, dtype=np.float64)
        model.damp_boundary(damp, nb, sp)
        out["damp_%s" % tag] = damp
        out["damp_%s_args" % tag] = np.array([shape[0], shape[1], nb, sp[0], sp[1]])
    damp32 = np.zeros((240, 240), dtype=np.float32)
    model.damp_boundary(damp32, 20, (10.0, 10.0))
    out["damp_a_f32"] = damp32

    # critical_dt / pad / vp setter on a namespace `self` ------------------------------------
    rng = np.random.default_rng(7)
    m = (1.0 / (1.5 + 2.5 * rng.random((12, 17))) ** 2).astype(np.float32)
    fake = types.SimpleNamespace(shape=(12, 17), nbpml=6, spacing=(10.0, 12.5), scale=1.0)
    fake.m = types.SimpleNamespace(data=np.zeros((24, 29), dtype=np.float32))
    fake.pad = lambda data: model.Model.pad(fake, data)
    model.Model.vp.fset(fake, m)                       # square slowness in, vp derived
    fake.vp = fake._vp
    out["cd_m"] = m
    out["cd_vp"] = np.asarray(fake._vp)
    out["cd_padded"] = fake.m.data.copy()
    out["cd_dt"] = np.array(model.Model.critical_dt.fget(fake))
    out["cd_args"] = np.array([6, 10.0, 12.5])

    # TimeAxis ----------------------------------------------------------------------------------
    ta = []
    for start, stop, step in [(0.0, 1000.0, 1.4), (0.0, 750.0, 2.0454), (0.0, 150.0, 0.1),
                              (5.0, 333.0, 0.77)]:
        t = source.TimeAxis(start=start, stop=stop, step=step)
        ta.append([start, stop, step, t.num, t.stop])
    out["timeaxis"] = np.array(ta)

    # Ricker --------------------------------------------------------------------------------------
    t = np.linspace(0.0, 1000.0, 716)
    out["ricker_t"] = t
    for tag, f0 in [("10hz", 0.010), ("25hz", 0.025), ("70hz", 0.07)]:
        out["ricker_%s" % tag] = source.RickerSource.wavelet(None, f0, t)
        out["ricker_%s_f0" % tag] = np.array(f0)

    np.savez_compressed(os.path.join(OUT, "seisgan_helpers.npz"), **out)
    print("wrote seisgan_helpers.npz with", len(out), "arrays")

    # plain torch expressions of prop() ---------------------------------------------------------
    import torch
    torch.manual_seed(1234)
    e = {}
    nt, ns, nr = 64, 6, 9
    obs = torch.randn(nt, ns, nr)
    pred = torch.randn(nt, ns, nr, requires_grad=True)
    cte = 0.3 * torch.randn(nt, ns, nr)
    # networks.py:5418-5419
    omax, _ = torch.abs(obs).max(dim=0, keepdim=True)
    obs_norm = obs / (omax.abs() + 1e-10)
    # networks.py:5434-5440, 5454-5455 (seeded shuffle + strided batch pick)
    idx = torch.randperm(ns)
    obs_norm_s = obs_norm[:, idx, :]
    cte_s = cte[:, idx, :]
    num_batches, it = 2, 0
    b_obs = obs_norm_s[:, it::num_batches]
    b_cte = cte_s[:, it::num_batches]
    b_pred = pred[:, idx, :][:, it::num_batches]
    # networks.py:5467-5476
    d = b_pred - b_cte
    dmax, _ = torch.abs(d).max(dim=0, keepdim=True)
    dn = d / (dmax.abs() + 1e-10)
    loss = torch.nn.L1Loss()(dn, b_obs)
    loss.backward()
    e.update(obs=obs.numpy(), pred=pred.detach().numpy(), cte=cte.numpy(), idx=idx.numpy(),
             obs_norm=obs_norm.numpy(), l1_loss=loss.detach().numpy(),
             l1_grad_pred=pred.grad.numpy(), num_batches=np.array(num_batches))
    # networks.py:5329-5332, 5492-5493 : z^2 ramp + water mask
    nz, nx = 11, 200
    grad = torch.randn(nz, nx)
    true = 1500.0 + 100.0 * torch.randint(0, 3, (1, 1, nz, nx)).float()
    g1 = torch.arange(nz) ** 2.0
    ss = torch.transpose(g1.tile((200, 1)), 0, 1)
    gc = grad * ss
    gc[(true[0, 0, :, :] == 1500)] = 0
    e.update(ac_grad=grad.numpy(), ac_true=true.numpy(), ac_grad_cond=gc.numpy())
    # networks.py:7808-7862 : flipud, mute rows 0:25, max-ratio rescale, rho x0.1
    nz, nx = 40, 30
    rngn = np.random.default_rng(5)
    gvp, gvs, grho = (rngn.standard_normal((nz, nx)).astype(np.float32) for _ in range(3))
    vpst, vsst, rhost = (np.abs(rngn.standard_normal((nz, nx))).astype(np.float32) * 1000
                         for _ in range(3))
    outs = []
    for gi, mi, fac in [(gvp, vpst, 1.0), (gvs, vsst, 1.0), (grho, rhost, 0.1)]:
        gg = np.flipud(gi).copy()
        gg[0:25, :] = 0.0
        rr = np.max(mi) / np.max(gg)
        outs.append((1.0 * torch.from_numpy(gg.copy()).float() * rr * fac).numpy())
    e.update(el_g=np.stack([gvp, gvs, grho]), el_m=np.stack([vpst, vsst, rhost]),
             el_g_cond=np.stack(outs))
    np.savez_compressed(os.path.join(OUT, "prop_expressions.npz"), **e)
    print("wrote prop_expressions.npz with", len(e), "arrays")


if __name__ == "__main__":
    main()
