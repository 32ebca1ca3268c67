"""Pins oracle/misfit.py to the reference's own expressions (models/networks.py:5418-5419,
5467-5476, 5491; seisgan/fwi/layers.py:176-178) evaluated by torch autograd on the CPU."""
import numpy as np
import pytest
import torch

from oracle import misfit as M


def _reference_expressions(pred, obs, direct):
    """The lines of networks.py:5467-5476 with their variable names shortened."""
    rcv_true_max, _ = torch.abs(obs).max(dim=0, keepdim=True)               # 5418
    rcv_true_norm = obs / (rcv_true_max.abs() + 1e-10)                      # 5419
    p = pred - direct                                                       # 5467
    p_max, _ = torch.abs(p).max(dim=0, keepdim=True)                        # 5468
    p_norm = p / (p_max.abs() + 1e-10)                                      # 5470
    loss = torch.nn.L1Loss()(p_norm, rcv_true_norm)                         # 5476
    return loss, rcv_true_norm


@pytest.mark.parametrize("shape", [(50, 3, 7), (131, 1, 1), (17, 2, 65)])
def test_l1_trace_norm_matches_torch_autograd(shape):
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(shape, generator=g, dtype=torch.float64, requires_grad=True)
    obs = torch.randn(shape, generator=g, dtype=torch.float64)
    direct = 0.3 * torch.randn(shape, generator=g, dtype=torch.float64)
    loss, obs_norm = _reference_expressions(pred, obs, direct)
    loss.backward()
    lo, adj = M.l1_trace_normalized(pred.detach().numpy(), obs_norm.numpy(), direct.numpy())
    assert abs(lo - loss.item()) <= 1e-14 * abs(loss.item()) + 1e-300
    assert np.abs(adj - pred.grad.numpy()).max() <= 1e-14 * np.abs(adj).max()
    assert np.array_equal(M.trace_normalize(obs.numpy()), obs_norm.numpy())


def test_dead_trace_and_exact_zero_residual():
    pred = np.zeros((9, 2, 2)); pred[:, 0, 0] = np.linspace(-1, 1, 9)
    obs = M.trace_normalize(pred.copy())
    lo, adj = M.l1_trace_normalized(pred, obs)
    assert lo == 0.0 and not np.any(adj)          # sign(0) = 0; a dead trace divides by 1e-10 only


def test_l2_half_matches_torch():
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(40, 3, 5, generator=g, dtype=torch.float64, requires_grad=True)
    obs = torch.randn(40, 3, 5, generator=g, dtype=torch.float64)
    res = pred - obs
    loss = 0.5 * (res * res).sum()                                          # layers.py:176-178
    loss.backward()
    lo, adj = M.l2_half(pred.detach().numpy(), obs.numpy())
    assert abs(lo - loss.item()) <= 1e-13 * loss.item()
    assert np.abs(adj - pred.grad.numpy()).max() <= 1e-14


def test_global_correlation_oracle_matches_torch_autograd():
    """-sum_traces <s,o>/(|s||o|) written with torch ops and differentiated by autograd."""
    import torch
    from oracle import misfit as M
    rng = np.random.default_rng(3)
    pred, obs = rng.standard_normal((40, 3, 5)), rng.standard_normal((40, 3, 5))
    obs[:, 1, 2] = 0.0                                   # a dead observed trace
    p = torch.tensor(pred, requires_grad=True)
    o = torch.tensor(obs)
    num = (p * o).sum(0)
    den = p.pow(2).sum(0).sqrt() * o.pow(2).sum(0).sqrt()
    loss = -(torch.where(den > 0, num / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))).sum()
    loss.backward()
    l, g = M.global_correlation(pred, obs)
    assert abs(l - float(loss)) <= 1e-13 * abs(float(loss))
    assert np.abs(g - p.grad.numpy()).max() <= 1e-14 and np.abs(g[:, 1, 2]).max() == 0.0
