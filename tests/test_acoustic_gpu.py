"""HIP acoustic propagator vs the CPU oracle (fp32) on identical seeded inputs.

Tolerances (fp32): receiver traces and snapshots use the same fmaf chain as the oracle and
are expected to agree to rounding of the last bit (asserted: rel-L2 <= 1e-6); gradients sum
products over time and shots in a different order: rel-L2 <= 2e-5.
"""
import numpy as np
import pytest
import torch

from cases import acoustic_case, rel_l2

pytestmark = pytest.mark.gpu

TOL_TRACE = 1e-6
TOL_GRAD = 2e-5


def _run_hip(case, gs=0, budget=None, need_f=True):
    from physicsbasedfwi2_amd import acoustic
    dev = torch.device("cuda:0")
    r = torch.tensor(case["r"], dtype=torch.float32, device=dev, requires_grad=True)
    f = torch.tensor(case["f"], dtype=torch.float32, device=dev, requires_grad=need_f)
    kw = {} if budget is None else {"snapshot_budget": budget}
    rec = acoustic.propagate(r, f, torch.tensor(case["q0"]), torch.tensor(case["q1"]),
                             torch.tensor(case["sc"]), torch.tensor(case["sw"]),
                             torch.tensor(case["rc"]), torch.tensor(case["rw"]),
                             case["c0"], case["c1"], shots_per_group=gs, **kw)
    return r, f, rec


@pytest.mark.parametrize("kw", [
    dict(),                                             # n1 multiple of 4 after padding? (66) no
    dict(n0=33, n1=47, nb=5, ns=3, nrec=11),            # ragged sizes, tail groups
    dict(n0=70, n1=300, nb=10, ns=2, nrec=40, nt=60),   # several tiles in x and z (LX=64)
    dict(ntap=4, ns=2, nsrc=2, nrec=9),                 # bilinear taps, two sources per shot
    dict(h=(10.0, 15.0), ntap=4),                       # anisotropic spacing
])
def test_forward_backward_parity(oracle32, kw):
    case = acoustic_case(seed=3, **kw)
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"],
                                    case["sw"], case["rc"], case["rw"], case["c0"], case["c1"],
                                    save=True)
    r, f, rec = _run_hip(case)
    rec_h = rec.detach().cpu().numpy()
    assert np.isfinite(rec_h).all()
    assert np.abs(rec_o).max() > 0
    assert rel_l2(rec_h, rec_o) <= TOL_TRACE
    rng = np.random.default_rng(11)
    g = (rng.standard_normal(rec_o.shape) * np.abs(rec_o).max()).astype(np.float32)
    rec.backward(torch.tensor(g, device=rec.device))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                     case["rc"], case["rw"], g, G_o, case["c0"], case["c1"])
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= TOL_GRAD
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= TOL_GRAD


def test_bitwise_traces(oracle32):
    """Same fmaf chain on both sides: report (and require) exact equality of the traces for the
    single-tap case."""
    case = acoustic_case(seed=5, n0=48, n1=64, nb=6, nt=120)
    rec_o = oracle32.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"],
                                      case["sw"], case["rc"], case["rw"], case["c0"], case["c1"])
    _, _, rec = _run_hip(case, need_f=False)
    diff = np.abs(rec.detach().cpu().numpy() - rec_o).max()
    print("max |hip - oracle| on traces:", diff)
    assert diff == 0.0


@pytest.mark.parametrize("gs", [2, 3])
def test_shot_groups(oracle32, gs):
    case = acoustic_case(seed=7, ns=5, nt=70)
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"],
                                    case["sw"], case["rc"], case["rw"], save=True)
    r, f, rec = _run_hip(case, gs=gs)
    assert rel_l2(rec.detach().cpu().numpy(), rec_o) <= TOL_TRACE
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=rec.device))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                     case["rc"], case["rw"], g, G_o)
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= TOL_GRAD
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= TOL_GRAD


def test_time_checkpointing_matches_resident_snapshots(oracle32):
    """Tiny snapshot budget forces checkpointed segments + re-propagation; the gradient must
    not change (same arithmetic, same order)."""
    case = acoustic_case(seed=9, nt=101, ns=2)
    r1, f1, rec1 = _run_hip(case)
    g = torch.sign(rec1.detach())
    rec1.backward(g)
    N0, N1 = case["shape"]
    step_bytes = 4 * 2 * N0 * ((N1 + 3) // 4 * 4)
    r2, f2, rec2 = _run_hip(case, budget=step_bytes * 2 * 13)     # 13-step segments
    rec2.backward(g)
    assert torch.equal(rec1, rec2)
    assert torch.equal(r1.grad, r2.grad)
    assert torch.equal(f1.grad, f2.grad)


def test_determinism():
    case = acoustic_case(seed=13, ns=3)
    outs = []
    for _ in range(2):
        r, f, rec = _run_hip(case)
        rec.backward(torch.ones_like(rec))
        outs.append((rec.detach().clone(), r.grad.clone(), f.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_zero_residual_zero_gradient():
    case = acoustic_case(seed=15)
    r, f, rec = _run_hip(case)
    rec.backward(torch.zeros_like(rec))
    assert float(r.grad.abs().max()) == 0.0
    assert float(f.grad.abs().max()) == 0.0


def test_deepwave_shim_matches_oracle(oracle32):
    """End-to-end through the deepwave-shaped API (metres, seconds, m/s), including the
    replicate-pad, the vp -> r chain rule and the L1 trace-normalised misfit of
    models/networks.py:5467-5476."""
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    from oracle import helpers as H
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    nz, nx, dx, dt, nt, P = 31, 52, 10.0, 0.001, 150, 12
    vp_np = (1500.0 + 2000.0 * rng.random((nz, nx))).astype(np.float32)
    ns, nr = 3, 20
    x_s = torch.zeros(ns, 1, 2)
    x_s[:, 0, 1] = torch.linspace(0, (nx - 1) * dx, ns)
    x_r = torch.zeros(ns, nr, 2)
    x_r[:, :, 1] = (torch.arange(nr).float() * 25.0)[None, :]
    wav = deepwave.wavelets.ricker(12.0, nt, dt, 1 / 12.0).reshape(-1, 1, 1).repeat(1, ns, 1)
    vp = torch.tensor(vp_np, device=dev, requires_grad=True)
    prop = deepwave.scalar.Propagator({"vp": vp}, dx, pml_width=P, absorbing="sponge")
    rec = prop(wav.to(dev), x_s.to(dev), x_r.to(dev), dt)
    obs = torch.tensor(rng.standard_normal((nt, ns, nr)).astype(np.float32), device=dev)
    pmax, _ = rec.abs().max(dim=0, keepdim=True)
    loss = torch.nn.L1Loss()(rec / (pmax + 1e-10), obs)
    loss.backward()
    # oracle composition in numpy/torch-CPU
    vpt = torch.tensor(vp_np, requires_grad=True)
    vpp = torch.nn.functional.pad(vpt[None, None], (P, P, P, P), mode="replicate")[0, 0]
    r_t = (vpp * (dt / dx)) ** 2
    N0, N1 = r_t.shape
    q0 = H.damp_profile_1d(N0, P, dx) * dx * dx / (2 * dt)
    q1 = H.damp_profile_1d(N1, P, dx) * dx * dx / (2 * dt)
    sc, sw = H.cell_taps(np.trunc(x_s[..., 0].numpy() / dx).astype(int) + P,
                         np.trunc(x_s[..., 1].numpy() / dx).astype(int) + P, N1)
    rc, rw = H.cell_taps(np.trunc(x_r[..., 0].numpy() / dx).astype(int) + P,
                         np.trunc(x_r[..., 1].numpy() / dx).astype(int) + P, N1)
    f_np = (wav.numpy() * dx * dx).astype(np.float32)
    r_np = r_t.detach().numpy()
    rec_o, G_o = oracle32.acoustic_forward(r_np, q0, q1, f_np, sc, sw, rc, rw, save=True)
    assert rel_l2(rec.detach().cpu().numpy(), rec_o) <= 1e-5
    rec_ot = torch.tensor(rec_o, requires_grad=True)
    pm, _ = rec_ot.abs().max(dim=0, keepdim=True)
    loss_o = torch.nn.L1Loss()(rec_ot / (pm + 1e-10), obs.cpu())
    loss_o.backward()
    gr_o, _ = oracle32.acoustic_backward(r_np, q0, q1, sc, sw, rc, rw, rec_ot.grad.numpy(), G_o,
                                         want_grad_f=False)
    r_t.backward(torch.tensor(gr_o))
    assert abs(float(loss) - float(loss_o)) <= 1e-5 * abs(float(loss_o))
    assert rel_l2(vp.grad.cpu().numpy(), vpt.grad.numpy()) <= 1e-4


@pytest.mark.parametrize("nw,edge", [(2, 0), (3, 0), (5, 0), (3, 9), (5, 9), (6, 12)])
def test_cluster_path_with_halo_handoff(oracle32, monkeypatch, nw, edge):
    """Force the LDS-resident cluster kernels to cut a shot into several row slabs so that the
    granule hand-off between workgroups is exercised; results must not change (bitwise traces).
    edge > 0: the first and last slab hold `edge` rows (the sponge), the others share the rest."""
    monkeypatch.setenv("MIFWI_AC_NW", str(nw))
    monkeypatch.setenv("MIFWI_AC_EDGE_ROWS", str(edge))
    case = acoustic_case(seed=17, n0=61, n1=83, nb=9, nt=140, ns=3, nrec=15)
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"],
                                    case["sw"], case["rc"], case["rw"], save=True)
    r, f, rec = _run_hip(case)
    assert np.abs(rec.detach().cpu().numpy() - rec_o).max() == 0.0
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=rec.device))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"],
                                     case["rc"], case["rw"], g, G_o)
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= TOL_GRAD
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= TOL_GRAD


def test_cluster_adjoint_source_paths_agree(oracle32, monkeypatch):
    """Adjoint sources of the single-launch time loop: a plain LDS read-add-write instead of an LDS float atomic
    (MIFWI_AC_ADJ_PLAIN=0 forces the atomic) - the same single rounding, the same bits where every tap has a cell of its
    own.  Taps that share a cell take turns in a fixed order: oracle parity, and bit-identical repeats."""
    monkeypatch.setenv("MIFWI_AC_NW", "3")
    case = acoustic_case(seed=59, n0=61, n1=83, nb=9, nt=120, ns=2, nrec=40)
    outs = []
    for flag in (None, "0"):
        if flag is not None:
            monkeypatch.setenv("MIFWI_AC_ADJ_PLAIN", flag)
        r, f, rec = _run_hip(case)
        rec.backward(torch.sign(rec.detach()))
        outs.append((r.grad.clone(), f.grad.clone()))
    assert float(outs[0][0].abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    monkeypatch.delenv("MIFWI_AC_ADJ_PLAIN")
    rc = case["rc"].copy()
    rc.reshape(2, -1)[:, 1] = rc.reshape(2, -1)[:, 0]             # two taps in one cell,
    rc.reshape(2, -1)[:, 5] = rc.reshape(2, -1)[:, 4]             # three in another: they take turns (lowest receiver
    rc.reshape(2, -1)[:, 6] = rc.reshape(2, -1)[:, 4]             # number first), a fixed order of additions
    case["rc"] = rc
    o = oracle32
    rec_o, G_o = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"], case["rc"],
                                    case["rw"], save=True)
    r, f, rec = _run_hip(case)
    assert np.abs(rec.detach().cpu().numpy() - rec_o).max() == 0.0
    g = np.sign(rec_o).astype(np.float32)
    rec.backward(torch.tensor(g, device=rec.device))
    gr_o, gf_o = o.acoustic_backward(case["r"], case["q0"], case["q1"], case["sc"], case["sw"], case["rc"], case["rw"],
                                     g, G_o)
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= TOL_GRAD
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= TOL_GRAD
    g1 = (r.grad.clone(), f.grad.clone())
    r, f, rec = _run_hip(case)                                    # and the same bits every time
    rec.backward(torch.tensor(g, device=rec.device))
    assert torch.equal(g1[0], r.grad) and torch.equal(g1[1], f.grad)


def test_cluster_and_per_step_paths_agree(monkeypatch):
    case = acoustic_case(seed=19, n0=70, n1=120, nb=10, nt=90, ns=2, nrec=21)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIFWI_AC_CLUSTER", flag)
        monkeypatch.setenv("MIFWI_AC_NW", "4" if flag == "1" else "0")
        r, f, rec = _run_hip(case)
        rec.backward(torch.sign(rec.detach()))
        outs.append((rec.detach().clone(), r.grad.clone(), f.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_l2(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= TOL_GRAD
    assert rel_l2(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()) <= TOL_GRAD


@pytest.mark.parametrize("family,ntap", [("1", 1), ("0", 1), ("0", 4)])
def test_born_operator_matches_oracle_and_transposes_the_gradient(oracle32, monkeypatch, family, ntap):
    """mifwi_acoustic_born (operators.py:168-207) in both kernel families: seismogram perturbation
    vs the oracle, and <J dr, g> = <dr, grad_r(g)> with the gradient of the autograd path."""
    from physicsbasedfwi2_amd import acoustic
    monkeypatch.setenv("MIFWI_AC_CLUSTER", family)
    if family == "1":
        monkeypatch.setenv("MIFWI_AC_NW", "3")
    case = acoustic_case(seed=23, n0=48, n1=70, nb=8, nt=110, ns=2, nrec=13, ntap=ntap)
    o = oracle32
    rng = np.random.default_rng(4)
    dr = (rng.standard_normal(case["r"].shape) * case["r"] * 0.05).astype(np.float32)
    rec_o, G = o.acoustic_forward(case["r"], case["q0"], case["q1"], case["f"], case["sc"], case["sw"],
                                  case["rc"], case["rw"], case["c0"], case["c1"], save=True)
    jdr_o = o.acoustic_born(case["r"], case["q0"], case["q1"], dr, G, case["rc"], case["rw"], case["c0"], case["c1"])
    dev = torch.device("cuda:0")
    args = [torch.tensor(case[k]) for k in ("q0", "q1", "sc", "sw", "rc", "rw")]
    r_t = torch.tensor(case["r"], dtype=torch.float32, device=dev)
    f_t = torch.tensor(case["f"], dtype=torch.float32, device=dev)
    rec, jdr = acoustic.born(r_t, f_t, torch.tensor(dr, device=dev), *args, case["c0"], case["c1"])
    assert rel_l2(rec.cpu().numpy(), rec_o) <= TOL_TRACE
    assert np.abs(jdr_o).max() > 0 and rel_l2(jdr.cpu().numpy(), jdr_o) <= 2e-6
    r2, f2, rec2 = _run_hip(case)
    g = torch.sign(jdr) + 0.5
    rec2.backward(g)
    lhs = float((jdr.double() * g.double()).sum())
    rhs = float((torch.tensor(dr, device=dev).double() * r2.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs))


def test_deepwave_shim_substeps_when_dt_exceeds_the_stability_limit(oracle32):
    """dt = 4 ms on a 10 m grid with 3.5 km/s needs an internal step of dt/ratio: the wavelet is
    resampled band-limited, the propagator runs ratio*nt steps and the traces are decimated."""
    import math
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    from physicsbasedfwi2_amd import profiles
    from physicsbasedfwi2_amd.compat.deepwave import scalar as shim
    from oracle import helpers as H
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(33)
    nz, nx, dx, dt, nt, P = 28, 40, 10.0, 0.004, 60, 10
    vp_np = (1500.0 + 2000.0 * rng.random((nz, nx))).astype(np.float32)
    ns, nr = 2, 12
    x_s = torch.zeros(ns, 1, 2); x_s[:, 0, 1] = torch.tensor([90.0, 300.0]); x_s[:, 0, 0] = 40.0
    x_r = torch.zeros(ns, nr, 2); x_r[:, :, 1] = (torch.arange(nr).float() * 30.0)[None, :]; x_r[:, :, 0] = 20.0
    wav = deepwave.wavelets.ricker(6.0, nt, dt, 1 / 6.0).reshape(-1, 1, 1).repeat(1, ns, 1)
    vp = torch.tensor(vp_np, device=dev, requires_grad=True)
    rec = deepwave.scalar.Propagator({"vp": vp}, dx, pml_width=P, absorbing="sponge")(wav.to(dev), x_s.to(dev), x_r.to(dev), dt)
    assert rec.shape == (nt, ns, nr)
    ratio = max(1, int(math.ceil(dt / (shim.CFL_SAFETY * profiles.scalar_cfl_limit([dx, dx], float(vp_np.max()))) - 1e-9)))
    assert ratio >= 2
    dti = dt / ratio
    vpp = np.pad(vp_np, P, mode="edge").astype(np.float64)
    r_np = ((vpp * (dti / dx)) ** 2).astype(np.float32)
    N0, N1 = r_np.shape
    q0 = H.damp_profile_1d(N0, P, dx) * dx * dx / (2 * dti)
    q1 = H.damp_profile_1d(N1, P, dx) * dx * dx / (2 * dti)
    sc, sw = H.cell_taps(np.trunc(x_s[..., 0].numpy() / dx).astype(int) + P,
                         np.trunc(x_s[..., 1].numpy() / dx).astype(int) + P, N1)
    rc, rw = H.cell_taps(np.trunc(x_r[..., 0].numpy() / dx).astype(int) + P,
                         np.trunc(x_r[..., 1].numpy() / dx).astype(int) + P, N1)
    f_up = shim._upsample(wav * (dx * dx), ratio).numpy().astype(np.float32)
    rec_o = oracle32.acoustic_forward(r_np, q0, q1, f_up, sc, sw, rc, rw)[::ratio]
    assert np.abs(rec_o).max() > 0 and rel_l2(rec.detach().cpu().numpy(), rec_o) <= 1e-5
    rec.square().sum().backward()
    assert torch.isfinite(vp.grad).all() and float(vp.grad.abs().max()) > 0


def _cpml_case(seed=5, n0=70, n1=90, w=10, nt=150, ns=3, nrec=24, ntap=1, h=(10.0, 10.0), nsrc=1):
    """acoustic_case with the sponge replaced by a W-cell C-PML: ab [2, n] = the a and b profiles of each axis."""
    from oracle import helpers as H
    c = acoustic_case(seed=seed, n0=n0, n1=n1, nb=w, nt=nt, ns=ns, nrec=nrec, ntap=ntap, h=h, nsrc=nsrc)
    N0, N1 = c["shape"]
    vmax = float(c["vp"].max())
    c["ab0"] = H.cpml_profiles(N0, w, h[0], c["s"], vmax, 0.02)[:2]
    c["ab1"] = H.cpml_profiles(N1, w, h[1], c["s"], vmax, 0.02)[:2]
    c["w"] = w
    # a source and a receiver INSIDE the layer as well (the reference puts both on the top row of the model)
    if ntap == 1:
        c["sc"][0, 0, 0] = 3 * N1 + N1 // 3
        c["rc"][0, 0, 0] = 2 * N1 + 4
    return c


def _run_cpml(c, budget=None, need_f=True):
    from physicsbasedfwi2_amd import acoustic
    dev = torch.device("cuda:0")
    r = torch.tensor(c["r"], dtype=torch.float32, device=dev, requires_grad=True)
    f = torch.tensor(c["f"], dtype=torch.float32, device=dev, requires_grad=need_f)
    kw = {} if budget is None else {"snapshot_budget": budget}
    rec = acoustic.propagate(r, f, torch.tensor(c["ab0"]), torch.tensor(c["ab1"]), torch.tensor(c["sc"]),
                             torch.tensor(c["sw"]), torch.tensor(c["rc"]), torch.tensor(c["rw"]), c["c0"], c["c1"],
                             cpml_width=c["w"], **kw)
    return r, f, rec


@pytest.mark.parametrize("kw", [
    dict(),
    dict(n0=41, n1=53, w=6, ns=2, nrec=9),                 # ragged sizes
    dict(n0=43, n1=57, w=7, ns=3, nrec=9),                 # N0 = 57 = 1 mod 4, odd width, odd shot count: the sub-buffers of the
                                                           # work area behind the layer's arrays (q1: float4 loads) stay 16-byte aligned
    dict(n0=60, n1=300, w=20, ns=2, nrec=40, nt=90),       # the reference's layer width, several tiles
    dict(ntap=4, nrec=9),                                  # bilinear taps
    dict(h=(10.0, 15.0)),                                  # anisotropic spacing: c0 != c1 in the layer's term
    dict(nsrc=2, nrec=12, n0=80, n1=120),                  # two sources per shot: the rescanning variants of the single-launch kernels
])
def test_cpml_forward_backward_parity(oracle32, kw):
    """Second-order C-PML (desc.cpml_width; what deepwave's pml_width is, networks.py:5408-5411) through the per-step
    kernels and their thin layer launches against oracle/acoustic_cpml.c: traces bit for bit, gradients <= 2e-5."""
    c = _cpml_case(**kw)
    o = oracle32
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rec_o, G_o = o.acoustic_cpml_forward(c["r"], c["ab0"], c["ab1"], c["f"], *geo, c["c0"], c["c1"], save=True)
    r, f, rec = _run_cpml(c)
    rec_h = rec.detach().cpu().numpy()
    assert np.abs(rec_o).max() > 0 and np.isfinite(rec_h).all()
    if c["sc"].shape[2] == 1:
        assert np.abs(rec_h - rec_o).max() == 0.0
    else:
        assert rel_l2(rec_h, rec_o) <= TOL_TRACE
    # the layer does something: the sponge-free scheme gives other traces
    z0, z1 = np.zeros(c["shape"][0]), np.zeros(c["shape"][1])
    plain = o.acoustic_forward(c["r"], z0, z1, c["f"], *geo, c["c0"], c["c1"])
    assert rel_l2(plain, rec_o) > 1e-6
    rng = np.random.default_rng(13)
    g = (rng.standard_normal(rec_o.shape) * np.abs(rec_o).max()).astype(np.float32)
    rec.backward(torch.tensor(g, device=rec.device))
    gr_o, gf_o = o.acoustic_cpml_backward(c["r"], c["ab0"], c["ab1"], *geo, g, G_o, c["c0"], c["c1"])
    assert rel_l2(r.grad.cpu().numpy(), gr_o) <= TOL_GRAD
    assert rel_l2(f.grad.cpu().numpy(), gf_o) <= TOL_GRAD


def test_cpml_time_checkpointing_carries_the_memory_variables():
    """The state a checkpoint keeps includes the layer's memory variables (layout.state_elems): a run cut into
    segments equals the resident one bit for bit."""
    c = _cpml_case(seed=9, nt=160)
    r1, f1, rec1 = _run_cpml(c)
    g = torch.sign(rec1.detach())
    rec1.backward(g)
    r2, f2, rec2 = _run_cpml(c, budget=1 << 20)
    rec2.backward(g)
    assert float(rec1.abs().max()) > 0 and torch.equal(rec1, rec2)
    assert torch.equal(r1.grad, r2.grad) and torch.equal(f1.grad, f2.grad)


def test_cpml_born_is_the_transpose_partner_of_the_gradient(oracle32):
    from physicsbasedfwi2_amd import acoustic
    c = _cpml_case(seed=21, ns=2, nt=120)
    dev = torch.device("cuda:0")
    t = lambda k: torch.tensor(c[k])
    r = torch.tensor(c["r"], dtype=torch.float32, device=dev)
    f = torch.tensor(c["f"], dtype=torch.float32, device=dev)
    rng = np.random.default_rng(2)
    dr = torch.tensor(c["r"] * 0.05 * rng.standard_normal(c["r"].shape), dtype=torch.float32, device=dev)
    rec, drec = acoustic.born(r, f, dr, t("ab0"), t("ab1"), t("sc"), t("sw"), t("rc"), t("rw"), c["c0"], c["c1"],
                              cpml_width=c["w"])
    rr, ff, rec2 = _run_cpml(c, need_f=False)
    assert torch.equal(rec, rec2.detach())
    g = torch.tensor(rng.standard_normal(tuple(rec.shape)).astype(np.float32), device=dev)
    rec2.backward(g)
    lhs, rhs = float((drec.double() * g.double()).sum()), float((rr.grad.double() * dr.double()).sum())
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs))
    # and J dr is the derivative of the forward map
    eps = 1e-2
    with torch.no_grad():
        kw = dict(cpml_width=c["w"])
        a = acoustic.propagate(r + eps * dr, f, t("ab0"), t("ab1"), t("sc"), t("sw"), t("rc"), t("rw"), c["c0"], c["c1"], **kw)
        b = acoustic.propagate(r - eps * dr, f, t("ab0"), t("ab1"), t("sc"), t("sw"), t("rc"), t("rw"), c["c0"], c["c1"], **kw)
    assert rel_l2(drec.cpu().numpy(), ((a - b) / (2 * eps)).cpu().numpy()) < 2e-2


def test_cpml_partial_lds_placements_give_the_same_bits(monkeypatch):
    """pml_place keeps as many of a slab's layer arrays in LDS as fit behind its planes - a prefix of a fixed list (6
    arrays forward, 10 adjoint; the axis-0 ones on the edge slabs only), the rest stays in global memory.  Which prefix
    depends on the grid; here the launch pretends to have 0 .. 80 KB less LDS (MIFWI_AC_PML_LDS_SHRINK_KB), which walks
    the cut through the list on edge and interior slabs alike, forward and adjoint - and takes the edge slabs from the
    own-group form of axis 0 (Psi / P, Q in LDS planes shaped like the field planes, e0 inside the update) to the generic
    one when the planes no longer fit: traces and gradients must be the same bits as with every array in global memory.  Offsets are added to the cell index, never to an LDS base pointer
    (a pointer biased below an LDS buffer leaves the LDS aperture)."""
    monkeypatch.setenv("MIFWI_AC_CLUSTER", "1")
    monkeypatch.setenv("MIFWI_AC_NW", "4")
    c = _cpml_case(seed=29, n0=88, n1=150, w=12, nt=70, ns=2, nrec=30)
    g = None
    outs = {}
    for shrink in (None, 0, "generic", "barrier", 3, 6, 9, 12, 16, 20, 26, 32, 40, 60, 80):
        # "generic": the edge slabs' layer of axis 0 through the thread maps and compact arrays (MIFWI_AC_PML_OWN=0) instead
        # of the own-group form with its LDS planes - which the larger shrinks switch off as well (its planes no longer fit);
        # "barrier": a barrier of its own behind the layer's last phase and the plain deal of the groups
        # (MIFWI_AC_PML_LATE_E=0) instead of the readers of e in slot 0, behind barrier A
        monkeypatch.setenv("MIFWI_AC_PML_OWN", "0" if shrink == "generic" else "1")
        monkeypatch.setenv("MIFWI_AC_PML_LATE_E", "0" if shrink == "barrier" else "1")
        if shrink is None:
            monkeypatch.setenv("MIFWI_AC_PML_LDS", "0")
        else:
            monkeypatch.setenv("MIFWI_AC_PML_LDS", "1")
            monkeypatch.setenv("MIFWI_AC_PML_LDS_SHRINK_KB", "0" if shrink in ("generic", "barrier") else str(shrink))
        r, f, rec = _run_cpml(c)
        if g is None:
            g = torch.sign(rec.detach()) + 0.25
        rec.backward(g)
        outs[shrink] = (rec.detach().clone(), r.grad.clone(), f.grad.clone())
    assert float(outs[None][0].abs().max()) > 0 and float(outs[None][1].abs().max()) > 0
    for shrink, o in outs.items():
        for a, b in zip(o, outs[None]):
            assert torch.equal(a, b), shrink


def test_cpml_single_launch_and_per_step_families_agree(oracle32, monkeypatch):
    """The C-PML inside the single-launch time loop (the layer's phases run by the slab's threads on the LDS-resident
    field, memory variables and the layer's term through the XCD's L2; edge slabs of W + 2 rows) against the per-step
    kernels with their thin layer launches and against the oracle: same cell functions, same bits."""
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    c = _cpml_case(seed=31, n0=100, n1=150, w=12, nt=140, ns=3, nrec=30)
    N0, N1 = c["shape"]
    pl = AcousticPlan(N0, N1, 140, 3, 1, 30, 1, 1.0, 1.0, 0, 0, 0, c["w"])
    assert pl.cluster_slabs() >= 3                             # the single-launch plan is what runs by default
    pl.close()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIFWI_AC_CLUSTER", flag)
        r, f, rec = _run_cpml(c)
        rec.backward(torch.sign(rec.detach()))
        outs.append((rec.detach().clone(), r.grad.clone(), f.grad.clone()))
    assert float(outs[0][0].abs().max()) > 0 and torch.equal(outs[0][0], outs[1][0])
    assert rel_l2(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 2e-5
    assert rel_l2(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()) <= 2e-5
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rec_o = oracle32.acoustic_cpml_forward(c["r"], c["ab0"], c["ab1"], c["f"], *geo, c["c0"], c["c1"])
    assert np.abs(outs[0][0].cpu().numpy() - rec_o).max() == 0.0
    # a forced slab count with thin interior slabs, and the checkpointed form (state copied out and in between calls)
    monkeypatch.setenv("MIFWI_AC_CLUSTER", "1")
    monkeypatch.setenv("MIFWI_AC_NW", "6")
    r2, f2, rec2 = _run_cpml(c, budget=1 << 21)
    rec2.backward(torch.sign(rec2.detach()))
    assert torch.equal(rec2.detach(), outs[0][0])
    assert rel_l2(r2.grad.cpu().numpy(), outs[1][1].cpu().numpy()) <= 2e-5


def test_geometry_cache_follows_tensor_identity_and_version():
    """propagate() keeps the device-resident geometry (and the validation of its cells: a host round trip) of tap tensors it
    has seen - while the very same device tensors are passed again, unmodified.  An in-place change makes a new one (and is
    validated again: an out-of-grid cell still raises); host tensors - which may alias numpy buffers - are never cached."""
    from physicsbasedfwi2_amd.acoustic import _Geometry
    from physicsbasedfwi2_amd._lib import MifwiError
    dev = torch.device("cuda:0")
    sc = torch.tensor([[[5]], [[9]]], dtype=torch.int32, device=dev)
    rc = torch.tensor([[[7], [8]], [[7], [8]]], dtype=torch.int32, device=dev)
    sw, rw = torch.ones(2, 1, 1, device=dev), torch.ones(2, 2, 1, device=dev)
    g1 = _Geometry.get(sc, sw, rc, rw, dev)
    assert _Geometry.get(sc, sw, rc, rw, dev) is g1
    g1.check_cells(100, "10x10")
    with pytest.raises(MifwiError):
        g1.check_cells(9, "3x3")
    rc[0, 0, 0] = 1000                                   # in place: the version counter moves
    g2 = _Geometry.get(sc, sw, rc, rw, dev)
    assert g2 is not g1
    with pytest.raises(MifwiError):
        g2.check_cells(100, "10x10")
    cpu = [t.cpu() for t in (sc, sw, rc, rw)]
    assert _Geometry.get(*cpu, dev) is not _Geometry.get(*cpu, dev)
