"""Gradient conditioning on the device the gradient lives on: the same assertions as
tests/test_golden_helpers.py::test_misfit_and_conditioning_expressions, tensors on cuda:0, against the
committed golden I/O pairs (tests/golden/prop_expressions.npz) and the host (numpy) form."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "prop_expressions.npz"))


def test_acoustic_gradient_conditioning_on_device(golden_dir):
    """networks.py:5329-5332, 5492-5493: z^2 ramp, water mask."""
    from physicsbasedfwi2_amd import conditioning as C
    g = _g(golden_dir)
    grad = torch.tensor(g["ac_grad"], device=DEV)
    out = C.condition_acoustic_gradient(grad, torch.tensor(g["ac_true"], device=DEV))
    assert out.device.type == "cuda"
    assert np.allclose(out.cpu().numpy(), g["ac_grad_cond"], rtol=1e-6)
    # the true model may stay on the host (the reference keeps it on another device, networks.py:5316-5333)
    out2 = C.condition_acoustic_gradient(grad, torch.tensor(g["ac_true"]))
    assert torch.equal(out, out2)
    assert torch.equal(grad, torch.tensor(g["ac_grad"], device=DEV))           # input untouched


def test_elastic_gradient_conditioning_on_device(golden_dir):
    """networks.py:7808-7862: flipud, mute rows 0:25, max-ratio rescale, rho x 0.1 - without leaving the
    device, equal to the host form and to the golden output."""
    from physicsbasedfwi2_amd import conditioning as C
    g = _g(golden_dir)
    gs = [torch.tensor(a, device=DEV) for a in g["el_g"]]
    ms = [torch.tensor(a, device=DEV) for a in g["el_m"]]
    dev = C.condition_elastic_gradients_on_device(*gs, *ms)
    host = C.condition_elastic_gradients(*g["el_g"], *g["el_m"])
    for d, h, ref in zip(dev, host, g["el_g_cond"]):
        assert d.device.type == "cuda"
        assert np.allclose(d.cpu().numpy(), ref, rtol=1e-6)
        assert torch.allclose(d.cpu(), h, rtol=1e-6, atol=0)
        assert float(d[:25].abs().max()) == 0.0


def test_misfit_expressions_and_gaussian_smoothing_on_device(golden_dir):
    """networks.py:5418-5419, 5467-5476 (torch mirror of the fused HIP misfit) and the sigma = 3 smoothing of
    networks.py:10526 with tensors on the GPU."""
    from scipy.ndimage import gaussian_filter
    from physicsbasedfwi2_amd import conditioning as C, misfit
    g = _g(golden_dir)
    obs = torch.tensor(g["obs"], device=DEV)
    assert torch.allclose(C.trace_normalize(obs).cpu(), torch.tensor(g["obs_norm"]), rtol=1e-6)
    idx, nb = torch.tensor(g["idx"], device=DEV), int(g["num_batches"])
    pred = torch.tensor(g["pred"], device=DEV, requires_grad=True)
    cte = torch.tensor(g["cte"], device=DEV)
    sel = lambda t: t[:, idx, :][:, 0::nb]
    loss = C.l1_trace_normalized(sel(pred), sel(C.trace_normalize(obs)), sel(cte))
    loss.backward()
    assert np.isclose(float(loss), float(g["l1_loss"]), rtol=1e-6)
    assert np.allclose(pred.grad.cpu().numpy(), g["l1_grad_pred"], rtol=1e-5, atol=1e-9)
    # the fused HIP kernel on the same batch
    p2 = sel(torch.tensor(g["pred"], device=DEV)).contiguous().requires_grad_(True)
    l2 = misfit.l1_trace_normalized(p2, sel(C.trace_normalize(obs)).contiguous(), sel(cte).contiguous())
    l2.backward()
    assert np.isclose(float(l2), float(g["l1_loss"]), rtol=1e-5)
    ref = torch.tensor(g["l1_grad_pred"], device=DEV)[:, idx, :][:, 0::nb]
    assert torch.allclose(p2.grad, ref, rtol=1e-4, atol=1e-8)
    rng = np.random.default_rng(9)
    a = rng.standard_normal((37, 53))
    sm = C.gaussian_smooth(torch.tensor(a, device=DEV), 3.0)
    assert np.abs(sm.cpu().numpy() - gaussian_filter(a, sigma=3.0)).max() < 1e-12


@pytest.mark.parametrize("sigma,mute,flip", [(0.0, 25, True), (3.0, 5, True), (1.3, 0, False), (7.5, 3, True)])
def test_fused_gradient_conditioning_kernel(golden_dir, sigma, mute, flip):
    """csrc/mifwi_gradient.hip against the reference's host expressions evaluated with numpy / scipy:
    networks.py:7808-7862 (flipud, mute, max-ratio, rho x 0.1), 10522-10540 (+ gaussian_filter) and a depth taper,
    on ragged sizes (tiles of 32 x 64, Gaussian radius up to 30 cells)."""
    from scipy.ndimage import gaussian_filter
    from physicsbasedfwi2_amd import conditioning as C
    rng = np.random.default_rng(21)
    nz, nx = 77, 203
    g = rng.standard_normal((3, nz, nx)).astype(np.float32)
    m = (np.abs(rng.standard_normal((3, nz, nx))) * 1000 + 500).astype(np.float32)
    w = (np.linspace(0.0, 2.0, nz) ** 2).astype(np.float32)
    fac = (1.0, 1.0, 0.1)
    ref = []
    for k in range(3):
        t = (g[k] * w[:, None]).astype(np.float32)
        t = np.flipud(t) if flip else t
        if sigma > 0:
            t = gaussian_filter(t, sigma=sigma)
        t = t.copy()
        t[0:mute] = 0.0
        ref.append(t * (np.max(m[k]) / np.max(t)) * fac[k])
    out = C.condition_gradients(torch.tensor(g, device=DEV), torch.tensor(m, device=DEV), torch.tensor(w), sigma, flip,
                                mute, fac)
    assert out.device.type == "cuda"
    o = out.cpu().numpy()
    for k in range(3):
        assert np.abs(o[k] - ref[k]).max() <= 2e-5 * np.abs(ref[k]).max(), k
        assert float(np.abs(o[k][:mute]).max() if mute else 0.0) == 0.0
    # the golden elastic pair (flipud, mute 25, ratio, rho x 0.1) through the same call
    gd = _g(golden_dir)
    out = C.condition_gradients(torch.tensor(gd["el_g"], device=DEV), torch.tensor(gd["el_m"], device=DEV), None, 0.0,
                                True, 25, fac)
    assert np.allclose(out.cpu().numpy(), gd["el_g_cond"], rtol=1e-6, atol=1e-6 * np.abs(gd["el_g_cond"]).max())
    # no models: plain factors; bad arguments fail loudly
    out = C.condition_gradients(torch.tensor(g, device=DEV), None, None, 0.0, False, 0, (2.0, 1.0, 1.0))
    assert torch.equal(out[0].cpu(), torch.tensor(g[0]) * 2.0) and torch.equal(out[1].cpu(), torch.tensor(g[1]))
    from physicsbasedfwi2_amd._lib import MifwiError
    with pytest.raises(MifwiError):
        C.condition_gradients(torch.tensor(g, device=DEV), sigma=9.0)
    with pytest.raises(MifwiError):
        C.condition_gradients(torch.tensor(g))
