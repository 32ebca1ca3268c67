"""HIP misfit kernels (through the C-ABI) vs the CPU oracle on identical seeded inputs.
fp32 kernel against the fp64 oracle: loss rel <= 2e-6, adjoint source rel-L2 <= 2e-6."""
import numpy as np
import pytest
import torch

from cases import rel_l2
from oracle import misfit as M

pytestmark = pytest.mark.gpu


def _data(shape, seed):
    rng = np.random.default_rng(seed)
    pred = rng.standard_normal(shape).astype(np.float32)
    direct = (0.3 * rng.standard_normal(shape)).astype(np.float32)
    obs = M.trace_normalize(rng.standard_normal(shape)).astype(np.float32)
    return pred, obs, direct


@pytest.mark.parametrize("shape,with_direct", [((50, 3, 7), True), ((131, 1, 1), False),
                                               ((17, 2, 65), True), ((2000, 4, 500), True),
                                               ((5, 1, 129), False)])
def test_l1_trace_norm_parity(shape, with_direct):
    from physicsbasedfwi2_amd import misfit
    pred, obs, direct = _data(shape, 11)
    dev = torch.device("cuda:0")
    p = torch.tensor(pred, device=dev, requires_grad=True)
    d = torch.tensor(direct, device=dev) if with_direct else None
    loss = misfit.l1_trace_normalized(p, torch.tensor(obs, device=dev), d)
    loss.backward()
    # oracle on the fp32 difference the kernel sees (pred - direct is formed in fp32 there)
    dd = (pred - direct) if with_direct else pred
    lo, adj = M.l1_trace_normalized(dd, obs)
    assert abs(loss.item() - lo) <= 2e-6 * abs(lo)
    assert rel_l2(p.grad.cpu().numpy(), adj) <= 2e-6


def test_matches_torch_expression_on_device():
    """Same numbers as the torch ops the reference runs on the GPU (conditioning.py mirror)."""
    from physicsbasedfwi2_amd import conditioning, misfit
    pred, obs, direct = _data((300, 5, 40), 13)
    dev = torch.device("cuda:0")
    a = torch.tensor(pred, device=dev, requires_grad=True)
    b = torch.tensor(pred, device=dev, requires_grad=True)
    o, d = torch.tensor(obs, device=dev), torch.tensor(direct, device=dev)
    l1 = misfit.l1_trace_normalized(a, o, d)
    l2 = conditioning.l1_trace_normalized(b, o, d)
    l1.backward(); l2.backward()
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l2.item())
    assert rel_l2(a.grad.cpu().numpy(), b.grad.cpu().numpy()) <= 2e-6


def test_l2_half_parity_and_scaling_of_upstream_gradient():
    from physicsbasedfwi2_amd import misfit
    pred, obs, _ = _data((123, 3, 37), 17)
    dev = torch.device("cuda:0")
    p = torch.tensor(pred, device=dev, requires_grad=True)
    loss = misfit.l2_half(p, torch.tensor(obs, device=dev))
    (2.0 * loss).backward()
    lo, adj = M.l2_half(pred, obs)
    assert abs(loss.item() - lo) <= 2e-6 * lo
    assert np.array_equal(p.grad.cpu().numpy(), 2.0 * (pred - obs))


def test_cpu_tensors_are_refused():
    from physicsbasedfwi2_amd import _lib, misfit
    with pytest.raises(_lib.MifwiError):
        misfit.l1_trace_normalized(torch.zeros(4, 2, 2, requires_grad=True), torch.zeros(4, 2, 2))


def test_second_backward_through_a_retained_graph():
    """The fused misfit keeps its adjoint source with the graph: two backward passes give the same gradient twice.
    The propagators free their snapshots in the first backward and say so when asked again."""
    from physicsbasedfwi2_amd import acoustic, misfit
    from physicsbasedfwi2_amd._lib import MifwiError
    from cases import acoustic_case
    dev = torch.device("cuda:0")
    pred = torch.randn(50, 3, 7, device=dev, requires_grad=True)
    obs = torch.randn(50, 3, 7, device=dev)
    loss = misfit.l2_half(pred, obs)
    (g1,) = torch.autograd.grad(loss, pred, retain_graph=True)
    (g2,) = torch.autograd.grad(loss, pred)
    assert torch.equal(g1, g2) and torch.allclose(g1, pred.detach() - obs)
    case = acoustic_case(seed=3, nt=40)
    r = torch.tensor(case["r"], dtype=torch.float32, device=dev, requires_grad=True)
    rec = acoustic.propagate(r, torch.tensor(case["f"], dtype=torch.float32, device=dev), torch.tensor(case["q0"]),
                             torch.tensor(case["q1"]), torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                             torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["c0"], case["c1"])
    rec.backward(torch.ones_like(rec), retain_graph=True)
    with pytest.raises(MifwiError, match="called twice"):
        rec.backward(torch.ones_like(rec))


def test_global_correlation_matches_oracle():
    from physicsbasedfwi2_amd import misfit
    rng = np.random.default_rng(5)
    nt, ns, nr = 301, 3, 70                             # ragged against the 64-trace / 16-slice tiling
    pred = (rng.standard_normal((nt, ns, nr)) * rng.random((1, ns, nr)) * 10).astype(np.float32)
    obs = (0.7 * pred + 0.5 * rng.standard_normal((nt, ns, nr))).astype(np.float32)
    obs[:, 0, 3] = 0.0
    pred[:, 2, 5] = 0.0
    dev = torch.device("cuda:0")
    p = torch.tensor(pred, device=dev, requires_grad=True)
    loss = misfit.global_correlation(p, torch.tensor(obs, device=dev))
    loss.backward()
    l, g = M.global_correlation(pred, obs)
    assert abs(float(loss.detach()) - l) <= 1e-5 * abs(l)
    assert rel_l2(p.grad.cpu().numpy(), g) <= 1e-5
    assert float(p.grad[:, 0, 3].abs().max()) == 0.0 and float(p.grad[:, 2, 5].abs().max()) == 0.0
    # scaling a trace does not change the misfit
    p2 = torch.tensor(pred * 3.0, device=dev)
    assert abs(float(misfit.global_correlation(p2, torch.tensor(obs, device=dev))) - l) <= 1e-5 * abs(l)
