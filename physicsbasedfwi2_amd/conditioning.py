"""The O(data) host expressions that surround the propagator call inside the reference's
``prop()`` methods: trace normalisation and L1 misfit (models/networks.py:5418-5419,
5467-5476), shot shuffle / strided mini-batch (5434-5440, 5454-5461), gradient conditioning
(5329-5332, 5492-5493; 7808-7862).  Plain torch / numpy; the fused HIP misfit lives in misfit.py, the fused HIP
gradient conditioning (:func:`condition_gradients`) in csrc/mifwi_gradient.hip.
"""
import numpy as np
import torch


def trace_normalize(d, eps=1e-10):
    """d / (max_t |d| + eps) per trace; d is [nt, nshot, nrec] (networks.py:5418-5419)."""
    dmax, _ = torch.abs(d).max(dim=0, keepdim=True)
    return d / (dmax.abs() + eps)


def l1_trace_normalized(pred, obs_norm, direct=None):
    """networks.py:5467-5476: subtract the direct wave, normalise per trace, L1 mean."""
    d = pred if direct is None else pred - direct
    return torch.nn.functional.l1_loss(trace_normalize(d), obs_norm)


def shuffle_and_pick(num_shots, num_batches, it=0, generator=None):
    """networks.py:5434-5461: idx = randperm(num_shots); batch `it` = idx[it::num_batches].
    Returns (idx, picked) so that callers can permute x_s, observed data and the direct wave
    identically (bit-exact given the torch RNG state)."""
    idx = torch.randperm(num_shots, generator=generator)
    return idx, idx[it::num_batches]


def condition_acoustic_gradient(grad, true_model, water_value=1500.0):
    """networks.py:5329-5332, 5492-5493: multiply row z by z^2, zero where the true model is
    water.  grad [nz,nx]; true_model [1,1,nz,nx]."""
    nz = grad.shape[0]
    ramp = (torch.arange(nz, device=grad.device, dtype=grad.dtype) ** 2.0)[:, None]
    out = grad * ramp
    out[(true_model[0, 0] == water_value).to(out.device)] = 0
    return out


def condition_elastic_gradients(g_vp, g_vs, g_rho, vp, vs, rho, mute_rows=25, rho_factor=0.1):
    """networks.py:7808-7862: flipud, zero rows 0:mute_rows, scale each by max(model)/max(grad);
    rho additionally x rho_factor.  numpy in, torch float tensors out (as the reference)."""
    outs = []
    for g, m, fac in ((g_vp, vp, 1.0), (g_vs, vs, 1.0), (g_rho, rho, rho_factor)):
        gg = np.flipud(np.asarray(g)).copy()
        gg[0:mute_rows, :] = 0.0
        r = np.max(m) / np.max(gg)
        outs.append(1.0 * torch.from_numpy(gg.copy()).float() * r * fac)
    return outs


def gaussian_smooth(g, sigma, truncate=4.0):
    """``scipy.ndimage.gaussian_filter(g, sigma)`` with its defaults (order 0, mode='reflect',
    truncate 4.0) as two separable 1-D correlations in torch, on whatever device ``g`` lives
    (networks.py:10526 smooths the RealData Vp gradient with sigma = 3).  ``g`` is [nz, nx]."""
    radius = int(truncate * float(sigma) + 0.5)
    x = torch.arange(-radius, radius + 1, device=g.device, dtype=torch.float64)
    k = torch.exp(-0.5 * (x / float(sigma)) ** 2)
    k = (k / k.sum()).to(g.dtype)
    out = g[None, None]
    for dim, pad in ((2, (0, 0, radius, radius)), (3, (radius, radius, 0, 0))):
        n = out.shape[dim]
        # scipy's 'reflect' is torch's 'symmetric' (edge sample repeated): build it by index
        idx = torch.arange(-radius, n + radius, device=g.device)
        idx = torch.where(idx < 0, -idx - 1, idx)
        idx = torch.where(idx >= n, 2 * n - 1 - idx, idx).clamp_(0, n - 1)
        ext = out.index_select(dim, idx)
        w = k.view(1, 1, -1, 1) if dim == 2 else k.view(1, 1, 1, -1)
        out = torch.nn.functional.conv2d(ext, w)
    return out[0, 0]


def condition_elastic_gradients_on_device(g_vp, g_vs, g_rho, vp, vs, rho, mute_rows=25, rho_factor=0.1):
    """:func:`condition_elastic_gradients` without leaving the device: torch tensors in (gradients
    as returned by ``Denise.get_fwi_gradients``, i.e. flipud-ed; models in the same orientation as
    the reference's ``vp, vs, rho``), torch tensors out."""
    outs = []
    for g, m, fac in ((g_vp, vp, 1.0), (g_vs, vs, 1.0), (g_rho, rho, rho_factor)):
        gg = torch.flip(torch.as_tensor(g), dims=(0,)).float().clone()
        gg[0:mute_rows, :] = 0.0
        outs.append(gg * (torch.as_tensor(m).max().to(gg.device) / gg.max()) * fac)
    return outs


def condition_gradients(grads, models=None, row_weight=None, sigma=0.0, flip=False, mute_rows=0, factors=None):
    """The whole post-processing of the model gradients in ONE library call on the device the gradients live on
    (csrc/mifwi_gradient.hip; no CPU path): depth taper -> scipy's Gaussian smoothing -> top mute -> max-ratio
    rescale, i.e. networks.py:7808-7862 (``flip=True, mute_rows=25, factors=(1, 1, 0.1)``) and 10522-10540
    (``flip=True, sigma=3, mute_rows=5``) without numpy round trips.

    grads [k, nz, nx] (k <= 4) CUDA tensor, as stored (``flip`` makes output row j read stored row nz-1-j);
    models [k, nz, nx] or None: ``out_k *= max(models_k) / max(out_k)``;  row_weight [nz] (indexed like the stored
    rows) or None;  factors: k floats.  Returns a new [k, nz, nx] tensor."""
    import ctypes
    from . import _lib
    if not grads.is_cuda:
        raise _lib.MifwiError("condition_gradients needs CUDA/HIP tensors (libmifwi has no CPU fallback)")
    lib = _lib.load()
    g = grads.detach().to(dtype=torch.float32).contiguous()
    if g.dim() != 3 or not 1 <= g.shape[0] <= 4:
        raise ValueError("grads must be [k, nz, nx] with k <= 4")
    k, nz, nx = g.shape
    dev = g.device
    m = None if models is None else torch.as_tensor(models).to(device=dev, dtype=torch.float32).contiguous()
    if m is not None and tuple(m.shape) != tuple(g.shape):
        raise ValueError("models must have the shape of grads")
    w = None if row_weight is None else torch.as_tensor(row_weight).to(device=dev, dtype=torch.float32).contiguous()
    if w is not None and tuple(w.shape) != (nz,):
        raise ValueError("row_weight must be [nz]")
    fac = None if factors is None else (ctypes.c_float * k)(*[float(v) for v in factors])
    out = torch.empty_like(g)
    work = torch.empty(lib.mifwi_gradient_condition_work_elems(k), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.mifwi_gradient_condition(dev.index or 0, _lib.ptr(g), _lib.ptr(m), _lib.ptr(out), k, nz, nx,
                                                _lib.ptr(w), float(sigma), int(bool(flip)), int(mute_rows),
                                                ctypes.cast(fac, ctypes.c_void_p) if fac is not None else None,
                                                _lib.ptr(work), torch.cuda.current_stream(dev).cuda_stream))
    return out
