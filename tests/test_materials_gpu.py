"""(Vp, Vs, rho) -> staggered material planes on the device (csrc/mifwi_materials.hip) against its definition, the torch
expression of physicsbasedfwi2_amd/elastic.py: the planes bit for bit (same operations in the same order), the chain rule
to fp32 round-off of a different summation order, and against central differences in float64."""
import numpy as np
import pytest
import torch

from physicsbasedfwi2_amd import elastic

pytestmark = pytest.mark.gpu


def _models(nz, nx, seed, water_rows):
    rng = np.random.default_rng(seed)
    vp = 1500.0 + 2500.0 * rng.random((nz, nx))
    vs = vp / (1.6 + 0.4 * rng.random((nz, nx)))
    rho = 1000.0 + 1500.0 * rng.random((nz, nx))
    vs[:water_rows] = 0.0                       # water: mu = 0, the harmonic mean's special case
    vs[nz // 2, nx // 3] = 0.0                  # and an isolated fluid cell
    return [torch.tensor(a, dtype=torch.float32) for a in (vp, vs, rho)]


@pytest.mark.parametrize("nz,nx,fs,water", [(37, 53, False, 0), (100, 300, True, 6), (5, 4, True, 1), (1, 9, False, 0), (9, 1, True, 0)])
def test_fused_materials_match_the_torch_expression(nz, nx, fs, water):
    dev = torch.device("cuda:0")
    dt, h = 2e-3, 20.0
    cpu = [t.clone().requires_grad_(True) for t in _models(nz, nx, 3, water)]
    gpu = [t.detach().to(dev).requires_grad_(True) for t in cpu]
    ref = elastic._staggered_materials_torch(*[t.to(dev) for t in cpu], dt, h, free_surface=fs)      # definition, on the device
    out = elastic.staggered_materials(*gpu, dt, h, free_surface=fs)
    assert out.shape == (5, nz, nx) and out.grad_fn is not None and type(out.grad_fn).__name__.startswith("_MaterialsFn")
    assert torch.equal(out, ref), [float((out[k] - ref[k]).abs().max()) for k in range(5)]
    # the CPU evaluation of the definition (what the oracle compositions of the tests use) agrees to round-off
    ref_cpu = elastic.staggered_materials(*cpu, dt, h, free_surface=fs)
    assert type(ref_cpu.grad_fn).__name__ != "_MaterialsFnBackward"
    assert float((out.detach().cpu() - ref_cpu.detach()).abs().max()) <= 2e-7 * float(ref_cpu.detach().abs().max())
    # chain rule: a random cotangent through both
    rng = np.random.default_rng(5)
    g = torch.tensor(rng.standard_normal((5, nz, nx)), dtype=torch.float32)
    out.backward(g.to(dev))
    ref_cpu.backward(g)
    for a, b, name in zip(gpu, cpu, ("vp", "vs", "rho")):
        err = float((a.grad.cpu() - b.grad).norm() / b.grad.norm())
        assert err <= 2e-6, (name, err)
    # same bits on a second evaluation (gather, no atomics)
    again = [t.detach().clone().requires_grad_(True) for t in gpu]
    elastic.staggered_materials(*again, dt, h, free_surface=fs).backward(g.to(dev))
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(again, gpu))


def test_fused_materials_directional_derivative():
    """<grad, d> from the fused chain rule against a central difference of the float64 definition."""
    dev = torch.device("cuda:0")
    nz, nx, dt, h = 24, 31, 1e-3, 10.0
    m = _models(nz, nx, 11, 3)
    rng = np.random.default_rng(12)
    d = [torch.tensor(rng.standard_normal((nz, nx)), dtype=torch.float64) * s for s in (30.0, 0.0, 20.0)]
    d[1] = torch.tensor(rng.standard_normal((nz, nx)), dtype=torch.float64) * 20.0 * (m[1] > 0)      # water stays water
    c = torch.tensor(rng.standard_normal((5, nz, nx)), dtype=torch.float64)
    f = lambda e: float((elastic.staggered_materials(*[a.double() + e * b for a, b in zip(m, d)], dt, h, free_surface=True) * c).sum())
    eps = 1e-3
    fd = (f(eps) - f(-eps)) / (2 * eps)
    gpu = [t.to(dev).requires_grad_(True) for t in m]
    (elastic.staggered_materials(*gpu, dt, h, free_surface=True) * c.float().to(dev)).sum().backward()
    an = sum(float((a.grad.double().cpu() * b).sum()) for a, b in zip(gpu, d))
    assert abs(an - fd) <= 2e-5 * abs(fd), (an, fd)


@pytest.mark.parametrize("nz,nx,pad", [(174, 500, 20), (33, 47, 6), (5, 3, 0), (1, 7, 4)])
def test_fused_acoustic_coefficients_match_the_torch_expression(nz, nx, pad):
    """vp -> r = (edge-replicated vp dt/h)^2 of the deepwave-shaped shim: the fused launch against pad, scale, square;
    the chain rule (layer folded into the edge cells) against autograd's, to the round-off of another summation order."""
    from physicsbasedfwi2_amd.compat.deepwave import scalar
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    vp0 = torch.tensor(1500.0 + 3000.0 * rng.random((nz, nx)), dtype=torch.float32, device=dev)
    c = 1e-3 / 10.0
    a = vp0.clone().requires_grad_(True)
    b = vp0.clone().requires_grad_(True)
    r = scalar._Coefficients.apply(a, pad, c)
    ref = (scalar._EdgePad.apply(b, pad) * c) ** 2
    assert r.shape == (nz + 2 * pad, nx + 2 * pad) and torch.equal(r, ref)
    g = torch.tensor(rng.standard_normal(tuple(r.shape)), dtype=torch.float32, device=dev)
    r.backward(g)
    ref.backward(g)
    assert float((a.grad - b.grad).norm() / b.grad.norm()) <= 1e-6
    a2 = vp0.clone().requires_grad_(True)
    scalar._Coefficients.apply(a2, pad, c).backward(g)
    assert torch.equal(a2.grad, a.grad)


@pytest.mark.parametrize("mode", [elastic.PARAM_VELOCITY, elastic.PARAM_IMPEDANCE, elastic.PARAM_LAME])
def test_gradient_parametrization_is_the_jacobian_of_the_change_of_variables(mode):
    """mifwi_elastic_gradient_parametrization (DENISE's INVMAT1, models/networks.py:11025) against float64 autograd of
    the change of variables itself: (Zp, Zs, rho) -> (Zp / rho, Zs / rho, rho) and (lambda, mu, rho) ->
    (sqrt((lambda + 2 mu) / rho), sqrt(mu / rho), rho), a random cotangent pulled back.  Water cells (Vs = 0) take no
    Vs term in the Lame form (the kernel's convention: the shear gradient is 0 there anyway - mu_xz = 0)."""
    dev = torch.device("cuda:0")
    vp, vs, rho = _models(41, 67, 11, 5)
    rng = np.random.default_rng(12)
    g = [torch.tensor(rng.standard_normal(vp.shape), dtype=torch.float32) for _ in range(3)]
    g[1][vs == 0] = 0.0
    out = elastic.gradient_parametrization([t.to(dev) for t in (vp, vs, rho)], [t.to(dev) for t in g], mode)
    P, Q, R = (t.double() for t in (vp, vs, rho))
    if mode == elastic.PARAM_VELOCITY:
        new = [P.clone().requires_grad_(True), Q.clone().requires_grad_(True), R.clone().requires_grad_(True)]
        old = new
    elif mode == elastic.PARAM_IMPEDANCE:
        new = [(R * P).requires_grad_(True), (R * Q).requires_grad_(True), R.clone().requires_grad_(True)]
        old = [new[0] / new[2], new[1] / new[2], new[2]]
    else:
        mu = R * Q * Q
        new = [(R * P * P - 2 * mu).requires_grad_(True), mu.requires_grad_(True), R.clone().requires_grad_(True)]
        wet = Q == 0
        shear = torch.where(wet, torch.zeros_like(mu), torch.sqrt(torch.where(wet, torch.ones_like(mu), new[1]) / new[2]))
        old = [torch.sqrt((new[0] + 2 * new[1]) / new[2]), shear, new[2]]
    sum((o * c.double()).sum() for o, c in zip(old, g)).backward()
    for k in range(3):
        want = new[k].grad
        got = out[k].cpu().double()
        assert float((got - want).abs().max()) <= 4e-6 * float(want.abs().max()), (mode, k)
    if mode == elastic.PARAM_VELOCITY:
        assert all(torch.equal(o.cpu(), c) for o, c in zip(out, g))
