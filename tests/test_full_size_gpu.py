"""BASELINE.json's grids at full size (C2 acoustic 174x500 + 20-cell sponge, 29 shots; C3 elastic
100x300, 32 shots) with a shortened time axis: the oracle where it finishes in seconds, otherwise
size-independent properties - both kernel families agree, run-to-run determinism, zero residual
=> zero gradient, adjoint dot-product identity through the C-ABI."""
import numpy as np
import pytest
import torch

from cases import acoustic_case, elastic_case, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _acoustic(case, need_grad=True):
    from physicsbasedfwi2_amd import acoustic
    dev = torch.device(DEV)
    r = torch.tensor(case["r"], dtype=torch.float32, device=dev, requires_grad=need_grad)
    f = torch.tensor(case["f"], dtype=torch.float32, device=dev, requires_grad=need_grad)
    rec = acoustic.propagate(r, f, torch.tensor(case["q0"]), torch.tensor(case["q1"]),
                             torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                             torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["c0"], case["c1"])
    return r, f, rec


def _elastic(case, need_grad=True):
    from physicsbasedfwi2_amd import elastic
    dev = torch.device(DEV)
    mat = torch.tensor(case["mat"], dtype=torch.float32, device=dev, requires_grad=need_grad)
    f = torch.tensor(case["f"], dtype=torch.float32, device=dev, requires_grad=need_grad)
    rvx, rvz = elastic.propagate(mat, f, torch.tensor(case["pz"]), torch.tensor(case["px"]),
                                 torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                                 torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["fw"])
    return mat, f, rvx, rvz


def test_plans_select_the_single_launch_family_on_the_reference_grids(monkeypatch):
    """A silent fall-back to one launch per step would keep every parity test green: check the
    selection itself (C2 with its sponge, C3, the 170x396 and 190x324 elastic grids)."""
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    assert AcousticPlan(214, 540, 100, 29, 1, 500, 1, 1.0, 1.0, 0).cluster_slabs() >= 1
    assert AcousticPlan(214, 540, 100, 29, 1, 500, 4, 1.0, 1.0, 0).cluster_slabs() == 0   # bilinear taps
    for nz, nx in ((100, 300), (150, 294), (170, 396), (190, 324)):
        pl = ElasticPlan(nz, nx, 100, 6, 1, 200, 1, 10, 0)
        assert pl.cluster_slabs(False) >= 1 and pl.cluster_slabs(True) >= 1, (nz, nx)
    assert ElasticPlan(400, 1974, 10, 2, 1, 200, 1, 10, 0).cluster_slabs(False) == 0
    monkeypatch.setenv("MIFWI_EL_CLUSTER", "0")
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", "0")
    pl = ElasticPlan(100, 300, 100, 6, 1, 200, 1, 10, 0)
    assert pl.cluster_slabs(False) == 0 and pl.cluster_slabs(True) == 0


def test_c2_acoustic_both_kernel_families_agree_and_are_deterministic(monkeypatch):
    case = acoustic_case(seed=41, n0=174, n1=500, nb=20, nt=300, ns=29, nrec=500)
    outs = []
    for flag in ("1", "1", "0"):
        monkeypatch.setenv("MIFWI_AC_CLUSTER", flag)
        r, f, rec = _acoustic(case)
        rec.backward(torch.sign(rec.detach()))
        outs.append((rec.detach().clone(), r.grad.clone(), f.grad.clone()))
    assert float(outs[0][0].abs().max()) > 0
    for a, b in zip(outs[0], outs[1]):                     # run-to-run: bit for bit
        assert torch.equal(a, b)
    assert torch.equal(outs[0][0], outs[2][0])             # cluster vs one launch per step: same traces
    assert rel_l2(outs[0][1].cpu().numpy(), outs[2][1].cpu().numpy()) <= 2e-5
    assert rel_l2(outs[0][2].cpu().numpy(), outs[2][2].cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("ns,nt", [(2, 300), (29, 120)])
def test_c2_acoustic_full_grid_vs_oracle(oracle32, ns, nt):
    """BASELINE config 2's grid (174x500 + 20-cell sponge = 214x540, 500 receivers per shot) through the
    plan the product picks by itself (single-launch time loop: many thin slabs for 2 shots, the 8-slab
    layout of the bench for 29) against the C oracle: traces bit for bit, gradients <= 2e-5."""
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    from oracle import helpers as H
    case = acoustic_case(seed=47, n0=174, n1=500, nb=20, nt=nt, ns=ns, nrec=500)
    # the acquisition of networks.py:5339-5355: sources and the 500 receivers on the top row of the model
    case["sc"], case["sw"] = H.cell_taps(np.full((ns, 1), 20), 20 + np.linspace(0, 499, ns).astype(int)[:, None], 540)
    case["rc"], case["rw"] = H.cell_taps(np.full((ns, 500), 20), np.tile(20 + np.arange(500), (ns, 1)), 540)
    assert AcousticPlan(214, 540, nt, ns, 1, 500, 1, 1.0, 1.0, 0).cluster_slabs() >= 1
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                    case["rc"], case["rw"], case["c0"], case["c1"], save=True)
    r, f, rec = _acoustic(case)
    rec_h = rec.detach().cpu().numpy()
    for s in range(ns):
        assert np.abs(rec_o[:, s]).max() > 0
    assert np.abs(rec_h - rec_o).max() == 0.0
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=DEV))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"], case["rc"],
                                     case["rw"], g, G_o, case["c0"], case["c1"])
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


@pytest.mark.parametrize("ns,nt", [(2, 200), (29, 100)])
def test_c2_acoustic_cpml_full_grid_vs_oracle(oracle32, monkeypatch, ns, nt):
    """BASELINE config 2's grid with `pml_width` as a PML (deepwave's meaning, networks.py:5408-5411): 174x500 + a 20-cell
    second-order C-PML = 214x540 through the single-launch time loop the plan picks.  With 29 shots (8 slabs per shot)
    the two edge slabs' planes leave LDS room for only a few of the layer's arrays - the mixed placement (pml_place) -
    against oracle/acoustic_cpml.c: traces bit for bit, gradients <= 2e-5; and the same with every layer array kept in
    global memory (MIFWI_AC_PML_LDS=0): identical bits."""
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    from oracle import helpers as H
    from test_acoustic_gpu import _cpml_case, _run_cpml
    c = _cpml_case(seed=53, n0=174, n1=500, w=20, nt=nt, ns=ns, nrec=500)
    c["sc"], c["sw"] = H.cell_taps(np.full((ns, 1), 20), 20 + np.linspace(0, 499, ns).astype(int)[:, None], 540)
    c["rc"], c["rw"] = H.cell_taps(np.full((ns, 500), 20), np.tile(20 + np.arange(500), (ns, 1)), 540)
    pl = AcousticPlan(214, 540, nt, ns, 1, 500, 1, 1.0, 1.0, 0, 0, 0, 20)
    assert pl.cluster_slabs() >= 3
    pl.close()
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rec_o, G_o = oracle32.acoustic_cpml_forward(c["r"], c["ab0"], c["ab1"], c["f"], *geo, c["c0"], c["c1"], save=True)
    g = np.sign(rec_o).astype(np.float32)
    gr_o, gf_o = oracle32.acoustic_cpml_backward(c["r"], c["ab0"], c["ab1"], *geo, g, G_o, c["c0"], c["c1"])
    outs = []
    for lds in ("1", "0"):
        monkeypatch.setenv("MIFWI_AC_PML_LDS", lds)
        r, f, rec = _run_cpml(c)
        rec_h = rec.detach().cpu().numpy()
        assert all(np.abs(rec_o[:, s]).max() > 0 for s in range(ns))
        assert np.abs(rec_h - rec_o).max() == 0.0
        rec.backward(torch.tensor(g, device=DEV))
        assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
        assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5
        outs.append((r.grad.clone(), f.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _seam_case(ns=2, nt=40, nz=1000, nx=3000, fw=10):
    """BASELINE config 5's grid and set-up (networks.py:9638, 9792-9811: 1000x3000, h = 30 m, dt = 2.5 ms,
    FREE_SURF = 1, C-PML on the other three sides) with a time axis the oracle finishes in seconds.  In 40
    steps a wave travels a few cells, so the acquisition is packed around the sources: shot 0 under the
    free surface (source at 180 m depth as in the reference, receivers ON the surface row and ten rows
    down), shot 1 in the bottom-left corner inside both C-PML strips."""
    from oracle import helpers as H
    rng = np.random.default_rng(7)
    h, dt = 30.0, 0.0025
    vp = 1500.0 + 3000.0 * np.linspace(0, 1, nz)[:, None] + 50.0 * rng.standard_normal((nz, nx))
    vs = vp / np.sqrt(3.0)
    rho = 310.0 * vp ** 0.25
    vs[:20] = 0.0; vp[:20] = 1500.0; rho[:20] = 1000.0
    mat = H.elastic_materials(vp, vs, rho, dt, h, free_surface=True)
    pz = H.cpml_profiles(nz, fw, h, dt, 1500.0, 5.0, lo=False)
    px = H.cpml_profiles(nx, fw, h, dt, 1500.0, 5.0)
    f = np.zeros((nt, ns, 1))
    f[:, :, 0] = (H.ricker_deepwave(20.0, nt, dt, 0.02) * 1e6)[:, None] * (1.0 + 0.25 * np.arange(ns))[None]
    xs = np.arange(1440, 1563, 3)
    sz, sx = [[6]], [[1500]]
    rz, rx = [np.r_[np.zeros(41, int), np.full(41, 10)]], [np.r_[xs, xs]]
    if ns > 1:
        sz.append([nz - 7]); sx.append([5])
        xc = np.arange(2, 84, 2)
        rz.append(np.r_[np.full(41, nz - 4), np.full(41, nz - 12)]); rx.append(np.r_[xc, xc])
    sc, sw = H.cell_taps(np.array(sz), np.array(sx), nx)
    rc, rw = H.cell_taps(np.array(rz), np.array(rx), nx)
    return dict(mat=mat, pz=pz, px=px, f=f, sc=sc, sw=sw, rc=rc, rw=rw, fw=fw)


def test_c5_seam_grid_free_surface_vs_oracle(oracle32):
    """1000x3000 with the free surface, one launch per half step (no LDS-resident plan at this size):
    seismograms bit for bit, the five material gradients and the source gradient <= 2e-5 against
    oracle/elastic.c."""
    from physicsbasedfwi2_amd import elastic
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    c = _seam_case()
    nt, ns = c["f"].shape[:2]
    pl = ElasticPlan(1000, 3000, nt, ns, 1, c["rc"].shape[1], 1, c["fw"], 0, 0, 1)
    assert pl.cluster_slabs(False) == 0 and pl.cluster_slabs(True) == 0
    o = oracle32
    ovx, ovz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], c["sc"], c["sw"], c["rc"], c["rw"],
                                    save=True, free_surface=1)
    dev = torch.device(DEV)
    mat = torch.tensor(c["mat"], dtype=torch.float32, device=dev, requires_grad=True)
    f = torch.tensor(c["f"], dtype=torch.float32, device=dev, requires_grad=True)
    rvx, rvz = elastic.propagate(mat, f, torch.tensor(c["pz"]), torch.tensor(c["px"]), torch.tensor(c["sc"]),
                                 torch.tensor(c["sw"]), torch.tensor(c["rc"]), torch.tensor(c["rw"]), c["fw"],
                                 free_surface=True)
    hx, hz = rvx.detach().cpu().numpy(), rvz.detach().cpu().numpy()
    for s in range(ns):                                   # every shot records something on both components
        assert np.abs(ovx[:, s]).max() > 0 and np.abs(ovz[:, s]).max() > 0
    assert np.abs(hx - ovx).max() == 0.0 and np.abs(hz - ovz).max() == 0.0
    gx, gz = np.sign(ovx).astype(np.float32), np.sign(ovz).astype(np.float32)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=dev), torch.tensor(gz, device=dev)])
    gm_o, gf_o = o.elastic_backward(c["mat"], c["pz"], c["px"], c["sc"], c["sw"], c["rc"], c["rw"], gx, gz, S,
                                    free_surface=1)
    del S
    gm_h = mat.grad.cpu().numpy()
    for k in range(5):
        assert np.abs(gm_o[k]).max() > 0 and rel_l2(gm_h[k], gm_o[k]) <= 2e-5, k
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


def test_c3_elastic_full_grid_vs_oracle(oracle32):
    case = elastic_case(seed=43, nz=100, nx=300, fw=10, ns=4, nrec=276, nt=160)
    o = oracle32
    ovx, ovz, S = o.elastic_forward(case["mat"], case["pz"], case["px"], case["f"], case["sc"],
                                    case["sw"], case["rc"], case["rw"], save=True, free_surface=0)
    mat, f, rvx, rvz = _elastic(case)
    hx, hz = rvx.detach().cpu().numpy(), rvz.detach().cpu().numpy()
    assert np.abs(hx - ovx).max() == 0.0 and np.abs(hz - ovz).max() == 0.0 and np.abs(ovx).max() > 0
    gx, gz = np.sign(ovx).astype(np.float32), np.sign(ovz).astype(np.float32)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=DEV), torch.tensor(gz, device=DEV)])
    gm_o, gf_o = o.elastic_backward(case["mat"], case["pz"], case["px"], case["sc"], case["sw"],
                                    case["rc"], case["rw"], gx, gz, S, free_surface=0)
    for k in range(5):
        assert rel_l2(mat.grad[k].cpu().numpy(), gm_o[k]) <= 2e-5
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


def test_c3_elastic_32_shots_families_agree_determinism_and_zero_residual(monkeypatch):
    case = elastic_case(seed=47, nz=100, nx=300, fw=10, ns=32, nrec=276, nt=200)
    outs = []
    for flag in ("1", "1", "0"):
        monkeypatch.setenv("MIFWI_EL_CLUSTER", flag)
        monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", flag)
        mat, f, rvx, rvz = _elastic(case)
        torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
        outs.append((rvx.detach().clone(), rvz.detach().clone(), mat.grad.clone(), f.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert rel_l2(outs[0][2].cpu().numpy(), outs[2][2].cpu().numpy()) <= 2e-5
    assert rel_l2(outs[0][3].cpu().numpy(), outs[2][3].cpu().numpy()) <= 2e-5
    # zero residual => zero gradient
    mat, f, rvx, rvz = _elastic(case)
    torch.autograd.backward([rvx, rvz], [torch.zeros_like(rvx), torch.zeros_like(rvz)])
    assert float(mat.grad.abs().max()) == 0.0 and float(f.grad.abs().max()) == 0.0


def test_c3_elastic_adjoint_dot_product_identity():
    """<J df, g> = <df, J^T g> for the source->seismogram map (linear in f) at full grid size:
    the adjoint kernels are the transpose of the forward kernels, whatever the size."""
    case = elastic_case(seed=53, nz=100, nx=300, fw=10, ns=8, nrec=276, nt=250)
    mat, f, rvx, rvz = _elastic(case)
    g = torch.Generator(device="cpu").manual_seed(1)
    gx = torch.randn(rvx.shape, generator=g).to(DEV)
    gz = torch.randn(rvz.shape, generator=g).to(DEV)
    torch.autograd.backward([rvx, rvz], [gx, gz])
    lhs = float((rvx.detach().double() * gx.double()).sum() + (rvz.detach().double() * gz.double()).sum())
    rhs = float((f.detach().double() * f.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs))


def test_interleaved_plans_of_different_grids():
    """forward A, forward B, backward A, backward B with different LDS footprints: the dynamic-LDS
    attribute of the cluster kernels is per function, so one plan must not shrink it for another."""
    big = acoustic_case(seed=61, n0=120, n1=300, nb=12, nt=80, ns=3, nrec=40)
    small = acoustic_case(seed=62, n0=30, n1=40, nb=6, nt=80, ns=2, nrec=9)
    ref = {}
    for name, case in (("big", big), ("small", small)):
        r, f, rec = _acoustic(case)
        rec.backward(torch.sign(rec.detach()))
        ref[name] = (rec.detach().clone(), r.grad.clone())
    ra, fa, reca = _acoustic(big)
    rb, fb, recb = _acoustic(small)
    reca.backward(torch.sign(reca.detach()))
    recb.backward(torch.sign(recb.detach()))
    assert torch.equal(reca.detach(), ref["big"][0]) and torch.equal(ra.grad, ref["big"][1])
    assert torch.equal(recb.detach(), ref["small"][0]) and torch.equal(rb.grad, ref["small"][1])
    ecase_a = elastic_case(seed=63, nz=90, nx=300, fw=10, ns=2, nrec=30, nt=60)
    ecase_b = elastic_case(seed=64, nz=40, nx=60, fw=6, ns=2, nrec=9, nt=60)
    ma, _, ax, az = _elastic(ecase_a)
    mb, _, bx, bz = _elastic(ecase_b)
    torch.autograd.backward([ax, az], [torch.sign(ax.detach()), torch.sign(az.detach())])
    torch.autograd.backward([bx, bz], [torch.sign(bx.detach()), torch.sign(bz.detach())])
    ma2, _, ax2, az2 = _elastic(ecase_a)
    torch.autograd.backward([ax2, az2], [torch.sign(ax2.detach()), torch.sign(az2.detach())])
    assert torch.equal(ax, ax2) and torch.equal(ma.grad, ma2.grad)


def test_c5_seam_sized_grid_checkpointed_equals_resident():
    """BASELINE's largest grid (1000x3000, networks.py:9638) on the one-launch-per-half-step family:
    time-checkpointed segments reproduce the resident-snapshot gradient bit for bit, and a zero
    residual gives a zero gradient."""
    from physicsbasedfwi2_amd import elastic
    nz, nx, ns, nt, fw = 1000, 3000, 2, 36, 10
    rng = np.random.default_rng(7)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import helpers as H
    vp = 1500.0 + 3000.0 * np.linspace(0, 1, nz)[:, None] + 50.0 * rng.standard_normal((nz, nx))
    vs = vp / np.sqrt(3.0); rho = 310.0 * vp ** 0.25
    vs[:20] = 0.0; vp[:20] = 1500.0; rho[:20] = 1000.0
    h, dt = 30.0, 0.0025
    dev = torch.device(DEV)
    prm = [torch.tensor(a.astype(np.float32), device=dev) for a in (vp, vs, rho)]
    mat0 = elastic.staggered_materials(*prm, dt, h)
    pz = torch.tensor(H.cpml_profiles(nz, fw, h, dt, 1500.0, 5.0))
    px = torch.tensor(H.cpml_profiles(nx, fw, h, dt, 1500.0, 5.0))
    f = torch.zeros(nt, ns, 1, device=dev)
    f[:, :, 0] = torch.tensor(H.ricker_deepwave(8.0, nt, dt, 0.02) * 1e6, dtype=torch.float32)[:, None]
    sc = torch.tensor([[[6 * nx + 700]], [[6 * nx + 2200]]], dtype=torch.int32)
    rx = np.arange(100, 2900, 7)
    rc = torch.tensor(np.tile((8 * nx + rx)[None, :, None], (ns, 1, 1)).astype(np.int32))
    ones = lambda t: torch.ones(t.shape)
    outs = []
    for budget in (96 << 30, 1 << 30):
        mat = mat0.clone().requires_grad_(True)
        rvx, rvz = elastic.propagate(mat, f, pz, px, sc, ones(sc), rc, ones(rc), fw, snapshot_budget=budget)
        torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
        outs.append((rvx.detach().clone(), mat.grad.clone()))
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][1].abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    mat = mat0.clone().requires_grad_(True)
    rvx, rvz = elastic.propagate(mat, f, pz, px, sc, ones(sc), rc, ones(rc), fw)
    torch.autograd.backward([rvx, rvz], [torch.zeros_like(rvx), torch.zeros_like(rvz)])
    assert float(mat.grad.abs().max()) == 0.0


def test_full_length_runs_agree_between_families(monkeypatch):
    """BASELINE's full time axes (C2: 2000 steps, 29 shots; C3: 3000 steps, 32 shots): ~10^5 halo
    hand-offs per workgroup in the single-launch kernels, compared sample by sample with the
    one-launch-per-step family."""
    case = acoustic_case(seed=81, n0=174, n1=500, nb=20, nt=2000, ns=29, nrec=500)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIFWI_AC_CLUSTER", flag)
        r, f, rec = _acoustic(case)
        rec.backward(torch.sign(rec.detach()))
        outs.append((rec.detach().clone(), r.grad.clone()))
        del r, f, rec
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0])
    assert rel_l2(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 5e-5
    del outs
    torch.cuda.empty_cache()
    ecase = elastic_case(seed=83, nz=100, nx=300, fw=10, ns=32, nrec=276, nt=3000)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIFWI_EL_CLUSTER", flag)
        monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", flag)
        mat, f, rvx, rvz = _elastic(ecase)
        torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
        outs.append((rvx.detach().clone(), rvz.detach().clone(), mat.grad.clone()))
        del mat, f, rvx, rvz
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel_l2(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()) <= 5e-5


def test_fused_steps_match_the_two_launch_form(monkeypatch):
    """MIFWI_EL_FUSED=1: V and S in one launch; MIFWI_EL_FUSED_ADJ=1: S^T and V^T in one launch (stencil operands
    staged in LDS, recomputed halo, ping-pong state; large grids).  Same arithmetic term by term: traces and
    gradients equal the two-launch kernels bit for bit, with the free surface, an odd number of steps, passes over
    shot subsets (copy back of a pass that ends in the second copy), shot groups sharing one accumulator set,
    checkpointed segments, bf16 snapshot planes and a grid wider and taller than one tile."""
    monkeypatch.setenv("MIFWI_EL_CLUSTER", "0")
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", "0")
    from physicsbasedfwi2_amd import elastic
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    case = elastic_case(seed=61, nz=70, nx=150, fw=8, ns=3, nrec=40, nt=75)
    big = 1 << 40
    outs = []
    for fused, adj, fsurf, budget, pass_shots, fmt, gs in (
            ("0", "0", 0, big, None, "f32", 0), ("1", "0", 0, big, None, "f32", 0), ("1", "1", 0, big, 2, "f32", 0),
            ("0", "0", 1, big, None, "f32", 2), ("0", "1", 1, big, None, "f32", 2), ("1", "1", 1, 3 << 20, 2, "f32", 2),
            ("0", "0", 1, big, None, "bf16", 0), ("1", "1", 1, big, 2, "bf16", 0), ("1", "1", 1, 3 << 20, None, "bf16", 0),
            # the column-walk adjoint (el_adj_walk): default chunking, one-iteration chunks, one chunk per column
            ("1", "2", 0, big, 2, "f32", 0), ("0", "2:16", 1, big, None, "f32", 2), ("1", "2:80", 1, 3 << 20, 2, "f32", 2),
            ("1", "2:32", 1, big, 2, "bf16", 0), ("0", "2", 1, 3 << 20, None, "bf16", 0)):
        monkeypatch.setenv("MIFWI_EL_FUSED", fused)
        monkeypatch.setenv("MIFWI_EL_FUSED_ADJ", adj.split(":")[0])
        if ":" in adj:
            monkeypatch.setenv("MIFWI_EL_WALK_ROWS", adj.split(":")[1])
        else:
            monkeypatch.delenv("MIFWI_EL_WALK_ROWS", raising=False)
        if pass_shots:
            monkeypatch.setenv("MIFWI_EL_FUSED_PASS_SHOTS", str(pass_shots))
            monkeypatch.setenv("MIFWI_EL_PASS_GROUPS", "1")
        else:
            monkeypatch.delenv("MIFWI_EL_FUSED_PASS_SHOTS", raising=False)
            monkeypatch.delenv("MIFWI_EL_PASS_GROUPS", raising=False)
        lay0 = ElasticPlan(70, 150, 75, 3, 1, 40, 1, 8, 0).layout
        mat = torch.tensor(case["mat"], dtype=torch.float32, device=DEV, requires_grad=True)
        f = torch.tensor(case["f"], dtype=torch.float32, device=DEV, requires_grad=True)
        rvx, rvz = elastic.propagate(mat, f, torch.tensor(case["pz"]), torch.tensor(case["px"]),
                                     torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                                     torch.tensor(case["rc"]), torch.tensor(case["rw"]), case["fw"],
                                     free_surface=fsurf, snapshot_budget=budget, snapshot_format=fmt,
                                     shots_per_group=gs)
        torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
        outs.append((lay0.work_forward_elems, lay0.work_backward_elems, rvx.detach(), rvz.detach(), mat.grad.clone(),
                     f.grad.clone()))
    assert outs[1][0] > outs[0][0] and outs[2][1] > outs[0][1]     # the fused plans carry a second copy of the state
    assert float(outs[0][2].abs().max()) > 0 and float(outs[0][4].abs().max()) > 0
    for a, b in ((0, 1), (0, 2), (3, 4), (3, 5), (6, 7), (6, 8), (0, 9), (3, 10), (3, 11), (6, 12), (6, 13)):
        for x, y in zip(outs[a][2:], outs[b][2:]):
            assert torch.equal(x, y), (a, b)
    assert not torch.equal(outs[3][4], outs[6][4])      # the bf16 planes were in use


def test_passes_over_shot_subsets_change_nothing(monkeypatch):
    """Large grids run the time loop over a few shots at a time (Infinity Cache residency).  Forced here on a
    small grid: traces bit for bit, gradients bit for bit for the same accumulator grouping."""
    monkeypatch.setenv("MIFWI_EL_CLUSTER", "0")
    monkeypatch.setenv("MIFWI_EL_CLUSTER_ADJ", "0")
    monkeypatch.setenv("MIFWI_AC_CLUSTER", "0")
    ce = elastic_case(seed=67, nz=60, nx=130, fw=8, ns=7, nrec=30, nt=60)
    ca = acoustic_case(seed=69, n0=60, n1=130, nb=8, nt=70, ns=7, nrec=30)
    outs = []
    for shots, groups in (("7", "4"), ("2", "1"), ("3", "2")):
        monkeypatch.setenv("MIFWI_EL_PASS_SHOTS", shots)
        monkeypatch.setenv("MIFWI_EL_PASS_GROUPS", groups)
        monkeypatch.setenv("MIFWI_AC_PASS_GROUPS", groups)
        mat, f, rvx, rvz = _elastic(ce)
        torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
        r, fa, rec = _acoustic(ca)
        rec.backward(torch.sign(rec.detach()))
        outs.append((rvx.detach(), rvz.detach(), mat.grad.clone(), f.grad.clone(),
                     rec.detach(), r.grad.clone(), fa.grad.clone()))
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][4].abs().max()) > 0
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)


def test_large_grids_sweep_the_time_range_a_few_shots_at_a_time():
    """Plans (no memory is allocated) for the SEAM-sized elastic grid and a large acoustic grid split their
    16 shots into Infinity-Cache-sized passes; grids whose shots all fit take them in one pass."""
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    pl = ElasticPlan(1000, 3000, 100, 16, 1, 300, 1, 10, 0)
    assert pl.cluster_slabs(False) == 0 and pl.cluster_slabs(True) == 0
    fwd, adj = pl.pass_sizes()
    assert fwd == 2 and adj == 1 and pl.layout.shots_per_group == 2      # 2 x 60 MB state + 60 MB materials
    small = ElasticPlan(350, 1700, 100, 8, 1, 300, 1, 10, 0)
    assert small.pass_sizes() == (8, small.layout.ngroups) and small.layout.shots_per_group == 4
    big_ac = AcousticPlan(1040, 3040, 100, 16, 1, 3000, 1, 1.0, 1.0, 0)
    assert big_ac.cluster_slabs() == 0
    f_ac, a_ac = big_ac.pass_sizes()
    assert 1 <= a_ac <= f_ac < big_ac.layout.ngroups
    c2 = AcousticPlan(214, 540, 100, 29, 1, 500, 1, 1.0, 1.0, 0)
    assert c2.pass_sizes() == (c2.layout.ngroups, c2.layout.ngroups)


# every other grid a prop() variant of the reference runs on, with the shot count of one
# of its batches and a shortened time axis: the plan the product picks by itself, traces bit for bit against the oracle
_ACOUSTIC_GRIDS = [(70, 70, 5, 70), (101, 101, 10, 101), (100, 200, 9, 200), (100, 250, 10, 250), (151, 200, 9, 200),
                   (151, 243, 9, 200), (151, 250, 6, 250), (151, 201, 4, 201), (201, 301, 5, 301)]


@pytest.mark.parametrize("n0,n1,ns,nrec", _ACOUSTIC_GRIDS, ids=lambda v: str(v))
def test_reference_acoustic_grids_vs_oracle(oracle32, n0, n1, ns, nrec):
    """networks.py:2711-16359 (the acoustic prop() variants): model n0 x n1 + 20-cell sponge, sources spread along the
    top row at non-integer spacing (linspace), one receiver per column."""
    from oracle import helpers as H
    nb, nt = 20, 150
    case = acoustic_case(seed=100 + n0 + n1, n0=n0, n1=n1, nb=nb, nt=nt, ns=ns, nrec=nrec)
    N1 = n1 + 2 * nb
    sx = np.floor(np.linspace(0, n1 - 1, ns)).astype(int)
    rx = np.floor(np.linspace(0, n1 - 1, nrec)).astype(int)
    case["sc"], case["sw"] = H.cell_taps(np.full((ns, 1), nb), nb + sx[:, None], N1)
    case["rc"], case["rw"] = H.cell_taps(np.full((ns, nrec), nb), np.tile(nb + rx, (ns, 1)), N1)
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                    case["rc"], case["rw"], case["c0"], case["c1"], save=True)
    r, f, rec = _acoustic(case)
    for s in range(ns):
        assert np.abs(rec_o[:, s]).max() > 0
    assert np.abs(rec.detach().cpu().numpy() - rec_o).max() == 0.0
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=DEV))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"], case["rc"],
                                     case["rw"], g, G_o, case["c0"], case["c1"])
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


@pytest.mark.parametrize("nz,nx,ns,nrec,fs", [(150, 294, 6, 223, False), (170, 396, 6, 370, False),
                                               (190, 324, 4, 301, True)], ids=lambda v: str(v))
def test_reference_elastic_grids_vs_oracle(oracle32, nz, nx, ns, nrec, fs):
    """networks.py:6216 (150x294), 8224 (170x396), 9637 (190x324 with FREE_SURF = 1): shots of one rank's share,
    a 10-cell C-PML frame, the receiver line of the reference's length."""
    from physicsbasedfwi2_amd import elastic
    case = elastic_case(seed=200 + nz, nz=nz, nx=nx, fw=10, ns=ns, nrec=nrec, nt=120, free_surface=fs)
    o = oracle32
    ovx, ovz, S = o.elastic_forward(case["mat"], case["pz"], case["px"], case["f"], case["sc"], case["sw"],
                                    case["rc"], case["rw"], save=True, free_surface=int(fs))
    dev = torch.device(DEV)
    mat = torch.tensor(case["mat"], dtype=torch.float32, device=dev, requires_grad=True)
    f = torch.tensor(case["f"], dtype=torch.float32, device=dev, requires_grad=True)
    rvx, rvz = elastic.propagate(mat, f, torch.tensor(case["pz"]), torch.tensor(case["px"]), torch.tensor(case["sc"]),
                                 torch.tensor(case["sw"]), torch.tensor(case["rc"]), torch.tensor(case["rw"]),
                                 case["fw"], free_surface=fs)
    hx, hz = rvx.detach().cpu().numpy(), rvz.detach().cpu().numpy()
    assert np.abs(ovx).max() > 0 and np.abs(ovz).max() > 0
    assert np.abs(hx - ovx).max() == 0.0 and np.abs(hz - ovz).max() == 0.0
    gx, gz = np.sign(ovx).astype(np.float32), np.sign(ovz).astype(np.float32)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=dev), torch.tensor(gz, device=dev)])
    gm_o, gf_o = o.elastic_backward(case["mat"], case["pz"], case["px"], case["sc"], case["sw"], case["rc"],
                                    case["rw"], gx, gz, S, free_surface=int(fs))
    for k in range(5):
        assert rel_l2(mat.grad[k].cpu().numpy(), gm_o[k]) <= 2e-5, k
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


@pytest.mark.parametrize("ns", [1, 3])
def test_c1_seisgan_shape_vs_oracle(oracle32, ns):
    """BASELINE config 1 at its own size: 200x200 + 20-cell sponge (240x240), bilinear (4-tap) source and 200
    bilinear receivers on a line (layers.py:81-89, 137-142), 1000 steps, 1 and 3 shots.  The taps are flattened
    to single-cell points so that the single-launch time loop serves it (neighbouring receivers then share
    cells: the turn-taking injection of the adjoint); against the C oracle with its native 4-tap points:
    traces <= 1e-6 (the taps of a point are summed in another order), gradients <= 2e-5."""
    from oracle import helpers as H
    from physicsbasedfwi2_amd import acoustic
    nt, nb = 1000, 20
    case = acoustic_case(seed=53, n0=200, n1=200, nb=nb, nt=nt, ns=ns, nrec=200, ntap=4)
    h = case["h"]
    sxy = np.zeros((ns, 1, 2)); sxy[:, 0, 0] = np.linspace(20.0, 1980.0, ns) if ns > 1 else 100.0; sxy[:, 0, 1] = 20.0
    rxy = np.zeros((ns, 200, 2)); rxy[:, :, 0] = np.linspace(0, 200 * 10.0, 200)[None]; rxy[:, :, 1] = 20.0
    case["sc"], case["sw"] = H.bilinear_taps(sxy, h, nb, (240, 240))
    case["rc"], case["rw"] = H.bilinear_taps(rxy, h, nb, (240, 240))
    dev = torch.device(DEV)
    r = torch.tensor(case["r"], dtype=torch.float32, device=dev, requires_grad=True)
    f = torch.tensor(case["f"], dtype=torch.float32, device=dev, requires_grad=True)
    geom = acoustic._Geometry(torch.tensor(case["sc"]), torch.tensor(case["sw"]), torch.tensor(case["rc"]),
                              torch.tensor(case["rw"]), dev)
    assert acoustic._flatten_taps_pays(r, f, geom, 1.0, 1.0, nb)      # the single-launch path is what runs
    rec = acoustic.propagate(r, f, torch.tensor(case["q0"]), torch.tensor(case["q1"]), torch.tensor(case["sc"]),
                             torch.tensor(case["sw"]), torch.tensor(case["rc"]), torch.tensor(case["rw"]),
                             case["c0"], case["c1"], edge_rows=nb)
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                    case["rc"], case["rw"], case["c0"], case["c1"], save=True)
    assert np.abs(rec_o).max() > 0 and rel_l2(rec.detach().cpu().numpy(), rec_o) <= 1e-6
    g = (rec_o / np.abs(rec_o).max()).astype(np.float32)
    rec.backward(torch.tensor(g, device=dev))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"], case["rc"],
                                     case["rw"], g, G_o, case["c0"], case["c1"])
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= 2e-5
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5


def test_marmousi2_10m_grid_vs_oracle(oracle32):
    """The 350x1700 grid of the bench's third line (10 m Marmousi-II) against the oracle:
    2 shots x 40 steps through the per-step kernels the plan picks at this size (fused V+S forward, S^T / V^T
    adjoint pair) - one shot under the top C-PML, one in the bottom-right corner: seismograms bit for bit, the
    five material gradients and the source gradient <= 2e-5."""
    from oracle import helpers as H
    from physicsbasedfwi2_amd import elastic
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    nz, nx, nt, ns, fw = 350, 1700, 40, 2, 10
    c = elastic_case(seed=59, nz=nz, nx=nx, fw=fw, nt=nt, ns=ns, nrec=82, h=10.0, dt=0.001, water=20, freq=25.0)
    xs = np.arange(790, 913, 3)
    sz, sx = [[4], [nz - 7]], [[850], [nx - 6]]
    xc = nx - 1 - np.arange(2, 84, 2)
    rz = [np.r_[np.full(41, 2), np.full(41, 12)], np.r_[np.full(41, nz - 4), np.full(41, nz - 12)]]
    rx = [np.r_[xs, xs], np.r_[xc, xc]]
    c["sc"], c["sw"] = H.cell_taps(np.array(sz), np.array(sx), nx)
    c["rc"], c["rw"] = H.cell_taps(np.array(rz), np.array(rx), nx)
    pl = ElasticPlan(nz, nx, nt, ns, 1, 82, 1, fw, 0)
    assert pl.cluster_slabs(False) == 0 and pl.cluster_slabs(True) == 0
    o = oracle32
    ovx, ovz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], c["sc"], c["sw"], c["rc"], c["rw"], save=True)
    mat, f, rvx, rvz = _elastic(c)
    hx, hz = rvx.detach().cpu().numpy(), rvz.detach().cpu().numpy()
    for s in range(ns):
        assert np.abs(ovx[:, s]).max() > 0 and np.abs(ovz[:, s]).max() > 0
    assert np.abs(hx - ovx).max() == 0.0 and np.abs(hz - ovz).max() == 0.0
    gx, gz = np.sign(ovx).astype(np.float32), np.sign(ovz).astype(np.float32)
    torch.autograd.backward([rvx, rvz], [torch.tensor(gx, device=DEV), torch.tensor(gz, device=DEV)])
    gm_o, gf_o = o.elastic_backward(c["mat"], c["pz"], c["px"], c["sc"], c["sw"], c["rc"], c["rw"], gx, gz, S)
    del S
    gm_h = mat.grad.cpu().numpy()
    for k in range(5):
        assert np.abs(gm_o[k]).max() > 0 and rel_l2(gm_h[k], gm_o[k]) <= 2e-5, k
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= 2e-5
