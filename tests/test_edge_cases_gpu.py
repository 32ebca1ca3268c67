"""Edge cases of the C-ABI calls: inactive taps, no receivers, a single time step, one shot, more
receivers than threads of a workgroup, sizes the stencil halo barely fits - each against the oracle
or an exact property.  Both kernel families where the plan offers both."""
import numpy as np
import pytest
import torch

from cases import acoustic_case, elastic_case, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ac(case, need_grad=True):
    from physicsbasedfwi2_amd import acoustic
    r = torch.tensor(case["r"], dtype=torch.float32, device=DEV, requires_grad=need_grad)
    f = torch.tensor(case["f"], dtype=torch.float32, device=DEV, requires_grad=need_grad)
    rec = acoustic.propagate(r, f, torch.tensor(case["q0"]), torch.tensor(case["q1"]),
                             torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                             torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["c0"], case["c1"])
    return r, f, rec


def _el(case, need_grad=True):
    from physicsbasedfwi2_amd import elastic
    mat = torch.tensor(case["mat"], dtype=torch.float32, device=DEV, requires_grad=need_grad)
    f = torch.tensor(case["f"], dtype=torch.float32, device=DEV, requires_grad=need_grad)
    rvx, rvz = elastic.propagate(mat, f, torch.tensor(case["pz"]), torch.tensor(case["px"]),
                                 torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                                 torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["fw"])
    return mat, f, rvx, rvz


@pytest.mark.parametrize("family", ["1", "0"])
def test_inactive_taps_read_zero_and_inject_nothing(oracle32, monkeypatch, family):
    monkeypatch.setenv("MIFWI_AC_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", family)
    case = acoustic_case(seed=71, n0=40, n1=56, nb=6, nt=60, ns=3, nsrc=2, nrec=9)
    case["sc"][1, 0, 0] = -1           # one dead source of shot 1
    case["rc"][:, 3, 0] = -1           # receiver 3 dead in every shot
    o = oracle32
    rec_o, G = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                  case["rc"], case["rw"], save=True)
    r, f, rec = _ac(case)
    assert np.array_equal(rec.detach().cpu().numpy(), rec_o) and not rec_o[:, :, 3].any()
    g = np.sign(rec_o).astype(np.float32) + 1.0     # non-zero adjoint source on the dead receiver too
    rec.backward(torch.tensor(g, device=DEV))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                     case["rc"], case["rw"], g, G)
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5 and not f.grad[:, 1, 0].any()
    ec = elastic_case(seed=72, nz=44, nx=60, fw=8, nt=60, ns=2, nsrc=2, nrec=9)
    ec["sc"][0, 1, 0] = -1
    ec["rc"][:, 5, 0] = -1
    ovx, ovz, S = o.elastic_forward(ec["mat"], ec["pz"], ec["px"], ec["f"], ec["sc"], ec["sw"], ec["rc"],
                                    ec["rw"], save=True)
    mat, ef, rvx, rvz = _el(ec)
    assert np.array_equal(rvx.detach().cpu().numpy(), ovx) and not ovx[:, :, 5].any()
    gx = np.ones_like(ovx); gz = np.ones_like(ovz)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=DEV), torch.tensor(gz, device=DEV)])
    gm_o, gf_o = o.elastic_backward(ec["mat"], ec["pz"], ec["px"], ec["sc"], ec["sw"], ec["rc"], ec["rw"],
                                    gx, gz, S)
    assert max(rel_l2(mat.grad[k].cpu().numpy(), gm_o[k]) for k in range(5)) <= 2e-5
    assert rel_l2(ef.grad.cpu().numpy(), gf_o) <= 2e-5 and not ef.grad[:, 0, 1].any()


@pytest.mark.parametrize("family", ["1", "0"])
def test_single_step_single_shot_and_smallest_grids(oracle32, monkeypatch, family):
    monkeypatch.setenv("MIFWI_AC_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", family)
    o = oracle32
    for kw in (dict(n0=5, n1=7, nb=1, nt=1, ns=1, nrec=2), dict(n0=9, n1=4, nb=2, nt=3, ns=1, nrec=1),
               dict(n0=33, n1=129, nb=4, nt=17, ns=1, nrec=5)):
        case = acoustic_case(seed=73, **kw)
        rec_o, G = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                      case["rc"], case["rw"], save=True)
        r, f, rec = _ac(case)
        assert np.array_equal(rec.detach().cpu().numpy(), rec_o), kw
        g = np.ones_like(rec_o)
        rec.backward(torch.tensor(g, device=DEV))
        gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                         case["rc"], case["rw"], g, G)
        assert np.abs(r.grad.cpu().numpy() - gr_o).max() <= 2e-5 * max(np.abs(gr_o).max(), 1e-30), kw
    ec = elastic_case(seed=74, nz=24, nx=20, fw=4, nt=1, ns=1, nrec=3, water=2)
    ovx, ovz = o.elastic_forward(ec["mat"], ec["pz"], ec["px"], ec["f"], ec["sc"], ec["sw"], ec["rc"], ec["rw"])
    mat, ef, rvx, rvz = _el(ec, need_grad=False)
    assert np.array_equal(rvx.cpu().numpy(), ovx) and np.array_equal(rvz.cpu().numpy(), ovz)


def test_more_receivers_than_threads_and_receivers_on_every_cell(oracle32):
    """1100 receivers per shot (> the 1024 / 512 threads of a cluster workgroup): the rescanning paths."""
    case = acoustic_case(seed=75, n0=30, n1=44, nb=4, nt=40, ns=2, nrec=1100)
    o = oracle32
    rec_o, G = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                  case["rc"], case["rw"], save=True)
    r, f, rec = _ac(case)
    assert np.array_equal(rec.detach().cpu().numpy(), rec_o)
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=DEV))
    gr_o, _ = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                  case["rc"], case["rw"], g, G)
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
    ec = elastic_case(seed=76, nz=30, nx=40, fw=4, nt=40, ns=2, nrec=700, water=3)
    rng = np.random.default_rng(1)
    ec["rc"] = rng.integers(0, 30 * 40, size=ec["rc"].shape).astype(np.int32)      # many per cell
    ovx, ovz, S = o.elastic_forward(ec["mat"], ec["pz"], ec["px"], ec["f"], ec["sc"], ec["sw"], ec["rc"],
                                    ec["rw"], save=True)
    mat, ef, rvx, rvz = _el(ec)
    assert np.array_equal(rvx.detach().cpu().numpy(), ovx) and np.array_equal(rvz.detach().cpu().numpy(), ovz)
    gx, gz = np.sign(ovx).astype(np.float32), np.sign(ovz).astype(np.float32)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=DEV), torch.tensor(gz, device=DEV)])
    gm_o, _ = o.elastic_backward(ec["mat"], ec["pz"], ec["px"], ec["sc"], ec["sw"], ec["rc"], ec["rw"], gx, gz, S)
    assert max(rel_l2(mat.grad[k].cpu().numpy(), gm_o[k]) for k in range(5)) <= 5e-5


def test_bad_arguments_fail_loudly():
    from physicsbasedfwi2_amd import _lib, acoustic
    case = acoustic_case(seed=77, n0=20, n1=24, nb=2, nt=5, ns=1, nrec=2)
    case["rc"][0, 0, 0] = case["shape"][0] * case["shape"][1] + 5          # outside the (padded) grid
    with pytest.raises(_lib.MifwiError):
        _ac(case)
    with pytest.raises(_lib.MifwiError):       # CPU tensors: no fallback
        acoustic.propagate(torch.zeros(8, 8), torch.zeros(3, 1, 1), torch.zeros(8), torch.zeros(8),
                           torch.zeros(1, 1, 1, dtype=torch.int32), torch.ones(1, 1, 1),
                           torch.zeros(1, 1, 1, dtype=torch.int32), torch.ones(1, 1, 1))


@pytest.mark.parametrize("family", ["1", "0"])
def test_no_receivers_or_no_sources(monkeypatch, family):
    """Empty point sets: shapes [ns, 0, ntap] must not fault and give empty / zero results."""
    monkeypatch.setenv("MIFWI_AC_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER", family)
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", family)
    case = acoustic_case(seed=91, n0=30, n1=40, nb=4, nt=20, ns=2, nrec=3)
    none_r = dict(case, rc=case["rc"][:, :0], rw=case["rw"][:, :0])
    r, f, rec = _ac(none_r, need_grad=False)
    assert tuple(rec.shape) == (20, 2, 0)
    none_s = dict(case, sc=case["sc"][:, :0], sw=case["sw"][:, :0], f=case["f"][:, :, :0])
    r, f, rec = _ac(none_s)
    assert tuple(rec.shape) == (20, 2, 3) and not rec.detach().cpu().numpy().any()
    rec.backward(torch.ones_like(rec))
    assert torch.isfinite(r.grad).all()
    ec = elastic_case(seed=92, nz=30, nx=40, fw=4, nt=20, ns=2, nrec=3, water=3)
    e0 = dict(ec, rc=ec["rc"][:, :0], rw=ec["rw"][:, :0])
    mat, ef, rvx, rvz = _el(e0, need_grad=False)
    assert tuple(rvx.shape) == (20, 2, 0)
    e1 = dict(ec, sc=ec["sc"][:, :0], sw=ec["sw"][:, :0], f=ec["f"][:, :, :0])
    mat, ef, rvx, rvz = _el(e1)
    assert not rvx.detach().cpu().numpy().any()
    torch.autograd.backward([rvx, rvz], [torch.ones_like(rvx), torch.ones_like(rvz)])
    assert torch.isfinite(mat.grad).all()
