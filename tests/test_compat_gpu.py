"""Drop-in boundary tests: the pyapi_denise-shaped and seisgan-shaped call protocols run on the
HIP propagator and agree with the CPU oracle composed the same way."""
import os

import numpy as np
import pytest
import torch

from cases import rel_l2
from oracle import helpers as H

pytestmark = pytest.mark.gpu


def _denise_setup(tmp_path):
    import physicsbasedfwi2_amd.compat.pyapi_denise as api
    rng = np.random.default_rng(3)
    nz, nx, dx = 60, 90, 20.0
    vp = (1800 + 1200 * rng.random((nz, nx))).astype(np.float32)
    vs = (vp / np.sqrt(3)).astype(np.float32)
    rho = (1900 + 300 * rng.random((nz, nx))).astype(np.float32)
    vs[:8] = 0; vp[:8] = 1500; rho[:8] = 1000
    d = api.Denise("/nonexistent", verbose=0)
    d.save_folder = str(tmp_path)
    d.set_paths()
    d.help()
    d.NPROCX, d.NPROCY, d.PHYSICS, d.ITERMAX = 6, 5, 1, 1
    d.TIME, d.DT, d.FREE_SURF, d.FW, d.FPML, d.DAMPING = 0.5, 0.002, 0, 10, 5.0, 1500.0
    xsrc = np.array([400.0, 1000.0, 1400.0])
    src = api.Sources(xsrc, 40.0 * xsrc / xsrc, 8.0)
    xrec = np.arange(300.0, 1500.0 + dx, 40.0)
    rec = api.Receivers(xrec, 460.0 * (xrec / xrec))
    return api, d, (vp, vs, rho), dx, src, rec


@pytest.mark.parametrize("order", [None, 4])
def test_denise_forward_matches_oracle(oracle32, tmp_path, order):
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    if order is None:
        assert d.FD_ORDER == 2       # the upstream default every reference prop() runs with (networks.py:10447 is a comment)
    else:
        d.FD_ORDER = order
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    sx, sy = d.forward(model, src, rec)
    assert sx.shape == (3, len(rec), 250) and len(d.get_shots(keys=["_y"])) == 3
    dt = 0.002
    mat = H.elastic_materials(vp, vs, rho, dt, dx)
    nz, nx = vp.shape
    pz, px = H.cpml_profiles(nz, 10, dx, dt, 1500.0, 5.0), H.cpml_profiles(nx, 10, dx, dt, 1500.0, 5.0)
    f = np.stack([api.ricker_denise(8.0, 250, dt)] * 3, axis=1)[:, :, None] * (dt / dx ** 2)
    iz = np.floor(src.y / dx + 0.5).astype(int) - 1
    ix = np.floor(src.x / dx + 0.5).astype(int) - 1
    sc, sw = H.cell_taps(iz[:, None], ix[:, None], nx)
    rz = np.floor(rec.y / dx + 0.5).astype(int) - 1
    rx = np.floor(rec.x / dx + 0.5).astype(int) - 1
    rc, rw = H.cell_taps(np.tile(rz, (3, 1)), np.tile(rx, (3, 1)), nx)
    ovx, ovz = oracle32.elastic_forward(mat, pz, px, f, sc, sw, rc, rw, fd_order=int(d.FD_ORDER))
    assert np.abs(ovz).max() > 0
    assert rel_l2(np.transpose(sy, (2, 0, 1)), ovz) < 1e-5
    assert rel_l2(np.transpose(sx, (2, 0, 1)), ovx) < 1e-5


def test_denise_grad_protocol_and_directional_derivative(tmp_path, monkeypatch):
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    monkeypatch.chdir(tmp_path)
    true = api.Model(np.flipud(vp * 1.03), np.flipud(vs * 0.98), np.flipud(rho), dx)
    ox, oy = d.forward(true, src, rec)
    d.set_observed(np.transpose(ox, (0, 2, 1)), np.transpose(oy, (0, 2, 1)))
    d.fwi_stages = []
    d.add_fwi_stage(fc_high=10, inv_rho_iter=10000)
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    d.grad(model, src, rec)
    loss = float(np.loadtxt("loss_curve_grad.out"))
    assert loss > 0 and np.isclose(loss, d.loss, rtol=1e-5)
    grads, names = d.get_fwi_gradients(["seis"], return_filenames=True)
    assert [n.split("_")[-1] for n in names] == ["rho.bin", "vp.bin", "vs.bin"]      # networks.py:7800-7802
    g_rho, g_vp, g_vs = (np.flipud(g) for g in grads)                                # caller's flipud
    assert g_vp.shape == vp.shape and np.isfinite(g_vp).all() and np.abs(g_vp).max() > 0
    # directional derivative along a smooth vp perturbation below the water layer
    dv = np.zeros_like(vp)
    dv[20:40, 30:60] = 30.0
    lin = float(np.sum(g_vp * dv))
    eps = 0.5
    lp = api.Denise(None, 0); lm = api.Denise(None, 0)
    vals = []
    for sgn in (+1, -1):
        dd = api.Denise(None, 0)
        for k in ("TIME", "DT", "FREE_SURF", "FW", "FPML", "DAMPING"):
            setattr(dd, k, getattr(d, k))
        dd.set_observed(np.transpose(ox, (0, 2, 1)), np.transpose(oy, (0, 2, 1)))
        dd.add_fwi_stage(fc_high=10, inv_rho_iter=10000)
        vals.append(dd.grad(api.Model(np.flipud(vp + sgn * eps * dv), np.flipud(vs), np.flipud(rho), dx),
                            src, rec))
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - lin) <= 0.03 * abs(fd), (fd, lin)
    # the post-processing of networks.py:7808-7862 on the device (api.conditioned_gradients) = the host expressions
    from physicsbasedfwi2_amd import conditioning as C
    host = C.condition_elastic_gradients(grads[1], grads[2], grads[0], vp, vs, rho)      # vp, vs, rho order
    devg = api.conditioned_gradients(d, vp, vs, rho)
    for a, b in zip(host, devg):
        assert b.is_cuda and torch.allclose(a, b.cpu(), rtol=2e-6, atol=0)
    # lnorm = 5: the global-correlation norm drives the same protocol (loss in [-2 nrec nshot, 0])
    d5 = api.Denise(None, 0)
    for k in ("TIME", "DT", "FREE_SURF", "FW", "FPML", "DAMPING"):
        setattr(d5, k, getattr(d, k))
    d5.set_observed(np.transpose(ox, (0, 2, 1)), np.transpose(oy, (0, 2, 1)))
    d5.add_fwi_stage(fc_high=10, inv_rho_iter=10000, lnorm=5)
    l5 = d5.grad(model, src, rec)
    assert -2.0 * len(rec) * len(src) <= l5 < 0 and np.abs(d5.get_fwi_gradients(["vp"])[0]).max() > 0
    d5.fwi_stages[-1]["lnorm"] = 7
    with pytest.raises(api.MifwiError if hasattr(api, "MifwiError") else Exception):
        d5.grad(model, src, rec)


def test_denise_impedance_and_lame_parameterisations(tmp_path, monkeypatch):
    """models/networks.py:10899-11110 (AutoElMarmousiMarZp22_Net.prop): the attribute assignments of that prop() re-typed,
    `d.INVMAT1 = 2` among them, then `d.grad`, `np.loadtxt('loss_curve_grad.out')`, `d.get_fwi_gradients(['seis'])` with
    grads[1] / grads[2] read as the "vp" / "vs" gradients - which in that parameterisation are the gradients with respect
    to Zp = rho Vp and Zs = rho Vs.  Checked: (1) composition - they equal the Vp / Vs / rho gradients pushed through the
    3 x 3 Jacobian of the change of variables in float64; (2) they ARE the derivative of the objective along a
    perturbation of Zp at fixed Zs, rho (central difference of d.grad's own loss); (3) the same for INVMAT1 = 3
    (lambda, mu, rho), water cells included (mu has no Vs term there)."""
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    monkeypatch.chdir(tmp_path)
    d.verbose = 0
    d.VPUPPERLIM, d.VPLOWERLIM, d.VSUPPERLIM, d.VSLOWERLIM = 4509.0, 1500.0, 2603.0, 0.0      # networks.py:11035-11040
    d.RHOUPPERLIM, d.RHOLOWERLIM = 2589.0, 1009.0
    d.SWS_TAPER_GRAD_HOR = 0
    true = api.Model(np.flipud(vp * 1.03), np.flipud(vs * 0.98), np.flipud(rho * 1.01), dx)
    ox, oy = d.forward(true, src, rec)
    obs = (np.transpose(ox, (0, 2, 1)), np.transpose(oy, (0, 2, 1)))
    d.set_observed(*obs)
    d.fwi_stages = []
    d.add_fwi_stage(fc_low=0.0, fc_high=10.0)
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    os.system("rm -rf loss_curve_grad.out")
    d.grad(model, src, rec)
    loss1 = float(np.loadtxt("loss_curve_grad.out"))
    assert loss1 > 0 and np.isclose(loss1, d.loss, rtol=1e-5)
    g_rho, g_vp, g_vs = (np.flipud(np.array(g)).astype(np.float64) for g in d.get_fwi_gradients(["seis"]))
    P, Q, R = vp.astype(np.float64), vs.astype(np.float64), rho.astype(np.float64)

    def run(mode, m=None):
        dd = api.Denise(None, 0)
        for k in ("TIME", "DT", "FREE_SURF", "FW", "FPML", "DAMPING", "PHYSICS", "ITERMAX"):
            setattr(dd, k, getattr(d, k))
        dd.INVMAT1 = mode
        dd.set_observed(*obs)
        dd.add_fwi_stage(fc_low=0.0, fc_high=10.0)
        loss = dd.grad(model if m is None else api.Model(*[np.flipud(a.astype(np.float32)) for a in m], dx), src, rec)
        grads, names = dd.get_fwi_gradients(["seis"], return_filenames=True)
        assert [n.split("_")[-1] for n in names] == ["rho.bin", "vp.bin", "vs.bin"]
        return loss, [np.flipud(np.array(g)).astype(np.float64) for g in grads]

    # (1) impedances: the same loss, gradients = J^T of (Zp, Zs, rho) -> (Vp, Vs, rho)
    loss2, (z_rho, z_p, z_s) = run(2)
    assert loss2 == d.loss                       # the same objective, bit for bit: only the gradients change variables
    assert rel_l2(z_p, g_vp / R) < 2e-6 and rel_l2(z_s, g_vs / R) < 2e-6
    assert rel_l2(z_rho, g_rho - (P * g_vp + Q * g_vs) / R) < 2e-5
    # (2) ... and the derivative of the objective along dZp at fixed Zs, rho: Vp changes by dZp / rho
    dz = np.zeros_like(P)
    dz[20:40, 30:60] = 6.0e4
    eps = 0.5
    lin = float(np.sum(z_p * dz))
    fd = (run(2, (P + eps * dz / R, Q, R))[0] - run(2, (P - eps * dz / R, Q, R))[0]) / (2 * eps)
    assert abs(fd - lin) <= 0.03 * abs(fd), (fd, lin)
    # density at fixed impedances: Vp = Zp / rho and Vs = Zs / rho move with it
    dr = np.zeros_like(P)
    dr[22:38, 32:58] = 40.0
    lin = float(np.sum(z_rho * dr))
    fd = (run(2, (P * R / (R + eps * dr), Q * R / (R + eps * dr), R + eps * dr))[0]
          - run(2, (P * R / (R - eps * dr), Q * R / (R - eps * dr), R - eps * dr))[0]) / (2 * eps)
    assert abs(fd - lin) <= 0.05 * abs(fd), (fd, lin)
    # (3) Lame parameters, water included (Vs = 0 on the first eight rows)
    _, (l_rho, l_lam, l_mu) = run(3)
    qs = np.where(Q == 0, 1.0, Q)
    assert rel_l2(l_lam, g_vp / (2 * R * P)) < 2e-6
    assert rel_l2(l_mu, g_vp / (R * P) + np.where(Q == 0, 0.0, g_vs / (2 * R * qs))) < 2e-6
    assert rel_l2(l_rho, g_rho - (P * g_vp + Q * g_vs) / (2 * R)) < 2e-5
    assert np.isfinite(l_mu).all() and np.abs(l_mu[:8]).max() > 0
    # what the shim does not serve is refused, not ignored
    with pytest.raises(api.MifwiError):
        d.INVMAT1 = 4
        d.grad(model, src, rec)
    d.INVMAT1 = 1
    with pytest.raises(AttributeError):
        d.INVMAT_1 = 2                           # a typo is not a parameter
    with pytest.raises(api.MifwiError):
        d.TIMEWIN = 1                            # changes the gradient, not built


def test_denise_free_surface_default_runs(oracle32, tmp_path):
    """pyapi default FREE_SURF=1 (what networks.py:7698-7731 leaves untouched, SEAM sets it at 9811)."""
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    d.FREE_SURF = 1
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    sx, sy = d.forward(model, src, rec)
    dt = 0.002
    nz, nx = vp.shape
    mat = H.elastic_materials(vp, vs, rho, dt, dx, free_surface=True)
    pz = H.cpml_profiles(nz, 10, dx, dt, 1500.0, 5.0, lo=False)
    px = H.cpml_profiles(nx, 10, dx, dt, 1500.0, 5.0)
    f = np.stack([api.ricker_denise(8.0, 250, dt)] * 3, axis=1)[:, :, None] * (dt / dx ** 2)
    iz = np.floor(src.y / dx + 0.5).astype(int) - 1
    ix = np.floor(src.x / dx + 0.5).astype(int) - 1
    sc, sw = H.cell_taps(iz[:, None], ix[:, None], nx)
    rz = np.floor(rec.y / dx + 0.5).astype(int) - 1
    rx = np.floor(rec.x / dx + 0.5).astype(int) - 1
    rc, rw = H.cell_taps(np.tile(rz, (3, 1)), np.tile(rx, (3, 1)), nx)
    ovx, ovz = oracle32.elastic_forward(mat, pz, px, f, sc, sw, rc, rw, free_surface=1, fd_order=int(d.FD_ORDER))
    assert rel_l2(np.transpose(sy, (2, 0, 1)), ovz) < 1e-5


def test_su_roundtrip(tmp_path):
    import physicsbasedfwi2_amd.compat.pyapi_denise as api
    a = np.random.default_rng(0).standard_normal((7, 33)).astype(np.float32)
    api.write_su(str(tmp_path / "x.su"), a, 0.002)
    b, dt = api.read_su(str(tmp_path / "x.su"))
    assert np.array_equal(a, b) and abs(dt - 0.002) < 1e-9


def test_seisgan_fwiloss_matches_oracle(oracle32):
    from physicsbasedfwi2_amd.compat.seisgan_fwi import FWIConfiguration, FWILoss
    rng = np.random.default_rng(5)
    nx, nz, nb = 50, 40, 10
    cfg = dict(origin=(0., 0.), shape=(nx, nz), spacing=(10., 10.), nbpml=nb, nshots=3,
               source_min_x=50., source_min_y=20., tn=400., t0=0., f0=0.015, nreceivers=24,
               rec_min_y=20., noise_percent=0.0)
    vp_true = 1.5 + 1.5 * rng.random((nx, nz))
    vp_0 = 1.5 + 1.5 * rng.random((nx, nz))
    m_true = (1.0 / vp_true ** 2).astype(np.float32)
    m0 = (1.0 / vp_0 ** 2).astype(np.float32)
    conf = FWIConfiguration(cfg, m_true)
    x = torch.tensor(m0[None, None], device="cuda:0", requires_grad=True)
    lossfn = FWILoss(conf)
    loss = lossfn(x)
    loss.backward()
    # --- oracle, composed by hand ---------------------------------------------------------------
    h = (10., 10.)
    dt = H.critical_dt(h, 1.0 / np.sqrt(m_true.min()))
    nt, _ = H.time_axis_num(0.0, 400.0, dt)
    assert nt == conf.nt and np.isclose(dt, conf.dt)
    t = np.linspace(0.0, dt * (nt - 1), nt)
    src = H.ricker_seisgan(0.015, t)
    N0, N1 = nx + 2 * nb, nz + 2 * nb
    d0, d1 = H.damp_profile_1d(N0, nb, 10.), H.damp_profile_1d(N1, nb, 10.)
    f = np.zeros((nt, 3, 1)); f[:nt - 2, :, 0] = (src[1:nt - 1] * 100.0)[:, None]
    sxy = np.zeros((3, 1, 2)); sxy[:, 0, 0] = np.linspace(50., 10. * nx - 50., 3); sxy[:, 0, 1] = 20.
    rxy = np.zeros((3, 24, 2)); rxy[:, :, 0] = np.linspace(0, 24 * 10., 24)[None]; rxy[:, :, 1] = 20.
    sc, sw = H.bilinear_taps(sxy, h, nb, (N0, N1))
    rc, rw = H.bilinear_taps(rxy, h, nb, (N0, N1))

    # The kernels and the fp32 oracle run the same fmaf chain, so they must be handed the same fp32 coefficients:
    # r, q and f rounded the way the shim rounds them (float32 division of a float32 constant by the padded
    # float32 model; float32 sponge profile and wavelet) - computed in float64 and rounded afterwards they differ
    # in the last bit and the loss moves by 1e-4 (round 2's bounds were 2e-4 / 2e-3 for that reason alone).
    _, q0_64, q1_64, _, _ = H.acoustic_coeffs(np.ones((N0, N1)), d0, d1, dt, h)
    q0, q1, f32 = conf.q0.numpy(), conf.q1.numpy(), conf.f.cpu().numpy()          # the shim's fp32 roundings ...
    assert np.allclose(q0, q0_64, rtol=1e-6, atol=0) and np.allclose(q1, q1_64, rtol=1e-6, atol=0)
    assert np.allclose(f32, f, rtol=1e-5, atol=1e-6 * np.abs(f).max())              # ... of the oracle-side formulas
    f = f32
    cst = np.float32(dt * dt / 100.0)

    def run(m):
        mp = H.pad_edge(m.astype(np.float32), nb)
        r = cst / mp
        rec, G = oracle32.acoustic_forward(r, q0, q1, f, sc, sw, rc, rw, 1.0, 1.0, save=True)
        syn = np.zeros_like(rec); syn[1:nt - 1] = rec[0:nt - 2]
        return r, mp, syn, G
    _, _, obs, _ = run(m_true)
    r, mp, syn, G = run(m0)
    assert rel_l2(conf.true_ds.cpu().numpy(), obs) < 1e-6
    res = syn - obs
    J = 0.5 * np.sum(res.astype(np.float64) ** 2)
    assert abs(float(loss) - J) <= 2e-5 * J
    g = np.zeros_like(res); g[0:nt - 2] = res[1:nt - 1]
    gr, _ = oracle32.acoustic_backward(r, q0, q1, sc, sw, rc, rw, g, G, want_grad_f=False)
    gm = (gr.astype(np.float64) * (-np.float64(cst) / mp.astype(np.float64) ** 2))[nb:-nb, nb:-nb]
    gm = gm / np.abs(gm).max()
    assert rel_l2(x.grad[0, 0].cpu().numpy(), gm) < 2e-5
    assert float(x.grad.abs().max()) == pytest.approx(1.0)


def test_denise_source_kinds_physics2_and_gradient_taper(tmp_path, monkeypatch):
    """QUELLART 3 (samples) fed with the QUELLART 1 wavelet reproduces it; QUELLART 6 runs;
    PHYSICS = 2 equals the elastic solver in a fluid; SWS_TAPER_GRAD_HOR multiplies every gradient
    by the documented depth window."""
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    monkeypatch.chdir(tmp_path)
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    sx1, sy1 = d.forward(model, src, rec)
    nt = sx1.shape[2]
    d.QUELLART = 3
    src3 = api.Sources(src.x, src.y, 8.0, wavelets=api.ricker_denise(8.0, nt, 0.002))
    sx3, sy3 = d.forward(model, src3, rec)
    assert np.array_equal(sx1, sx3) and np.array_equal(sy1, sy3)
    d.QUELLART, d.FC_SPIKE_1, d.FC_SPIKE_2 = 6, 3.0, 10.0
    sx6, _ = d.forward(model, src, rec)
    assert np.isfinite(sx6).all() and np.abs(sx6).max() > 0
    # acoustic = fluid
    d.QUELLART, d.PHYSICS = 1, 2
    ax, ay = d.forward(model, src, rec)
    d.PHYSICS = 1
    fluid = api.Model(np.flipud(vp), np.zeros_like(vs), np.flipud(rho), dx)
    fx, fy = d.forward(fluid, src, rec)
    assert np.array_equal(ax, fx) and np.array_equal(ay, fy)
    # gradient taper
    d.set_observed(0.9 * np.transpose(sx1, (0, 2, 1)), 0.9 * np.transpose(sy1, (0, 2, 1)))
    d.grad(model, src, rec)
    g0 = d.get_fwi_gradients(["seis"])
    d.SWS_TAPER_GRAD_HOR, d.EXP_TAPER_GRAD_HOR, d.GRADT3, d.GRADT4 = 1, 2.0, 50, 58
    d.grad(model, src, rec)
    g1 = d.get_fwi_gradients(["seis"])
    w = api.gradient_taper(vp.shape[0], dx, 21, 25, 50, 58, 2.0)[:, None]
    for a, b in zip(g0, g1):
        assert np.allclose(np.flipud(b), np.flipud(a) * w, rtol=1e-6, atol=0)
    assert not np.any(np.flipud(g1[1])[:21]) and not np.any(np.flipud(g1[1])[57:])


def test_denise_point_force_and_adjoint_source_components(oracle32, tmp_path, monkeypatch):
    """QUELLTYP 3 (vertical point force): seismograms equal the oracle fed with wavelet * dt/(h^2 rho) at the
    source node.  QUELLTYPB (DENISE's adjoint-source type; networks.py:10452 sets 2) selects the components
    of the misfit: 2 = y only, 3 = x only, and the two losses add up to QUELLTYPB = 1."""
    api, d, (vp, vs, rho), dx, src, rec = _denise_setup(tmp_path)
    monkeypatch.chdir(tmp_path)
    from oracle import helpers as H
    model = api.Model(np.flipud(vp), np.flipud(vs), np.flipud(rho), dx)
    ex, ey = d.forward(model, src, rec)
    d.QUELLTYP = 3
    sx, sy = d.forward(model, src, rec)
    assert np.abs(sy).max() > 0 and rel_l2(sy / np.abs(sy).max(), ey / np.abs(ey).max()) > 0.1
    nt = sx.shape[2]
    dt, nz, nx = 0.002, vp.shape[0], vp.shape[1]
    mat = H.elastic_materials(vp, vs, rho, dt, dx)
    pz, px = H.cpml_profiles(nz, 10, dx, dt, 1500.0, 5.0), H.cpml_profiles(nx, 10, dx, dt, 1500.0, 5.0)
    ix, iz = np.rint(src.x / dx).astype(int) - 1, np.rint(src.y / dx).astype(int) - 1
    jx, jz = np.rint(rec.x / dx).astype(int) - 1, np.rint(rec.y / dx).astype(int) - 1
    sc, sw = H.cell_taps(iz[:, None], ix[:, None], nx)
    rc, rw = H.cell_taps(np.tile(jz, (3, 1)), np.tile(jx, (3, 1)), nx)
    w = api.ricker_denise(8.0, nt, dt).astype(np.float64)
    f = np.stack([w * mat[4][iz[k], ix[k]] / dx for k in range(3)], axis=1)[:, :, None].astype(np.float32)
    ovx, ovz = oracle32.elastic_forward(mat, pz, px, f, sc, sw, rc, rw, source_type=2, fd_order=int(d.FD_ORDER))
    assert rel_l2(np.transpose(sy, (2, 0, 1)), ovz) <= 2e-5 and rel_l2(np.transpose(sx, (2, 0, 1)), ovx) <= 2e-5
    # adjoint-source components
    d.set_observed(0.8 * np.transpose(sx, (0, 2, 1)), 0.8 * np.transpose(sy, (0, 2, 1)))
    losses, grads = {}, {}
    for b in (1, 2, 3):
        d.QUELLTYPB = b
        losses[b] = d.grad(model, src, rec)
        grads[b] = d.get_fwi_gradients(["seis"])
    assert losses[2] > 0 and losses[3] > 0
    assert abs(losses[2] + losses[3] - losses[1]) <= 1e-5 * losses[1]
    for k in range(3):
        assert np.allclose(grads[2][k] + grads[3][k], grads[1][k], rtol=2e-4, atol=1e-6 * np.abs(grads[1][k]).max())
    d.QUELLTYPB = 4
    with pytest.raises(Exception, match="QUELLTYPB"):
        d.grad(model, src, rec)
    # pressure seismograms and pressure adjoint sources (SEISMO 4, QUELLTYPB 4); the velocities do not change
    d.QUELLTYP, d.SEISMO = 1, 4
    px_, py_ = d.forward(model, src, rec)
    assert np.array_equal(px_, ex) and np.array_equal(py_, ey)
    sp = np.stack(d.get_shots(keys=["_p"]))
    assert sp.shape == ey.shape and np.abs(sp).max() > 0
    d.set_observed(np.transpose(ex, (0, 2, 1)), np.transpose(ey, (0, 2, 1)), p=0.7 * np.transpose(sp, (0, 2, 1)))
    lp = d.grad(model, src, rec)
    assert abs(lp - 0.5 * 0.09 * float((sp.astype(np.float64) ** 2).sum())) <= 1e-4 * lp
    gp = d.get_fwi_gradients(["seis"])
    assert all(np.isfinite(a).all() for a in gp) and np.abs(gp[1]).max() > 0


# ------------------------------------------------------------------ deepwave shim, C-PML mode --
DEV = "cuda:0"
def _deepwave_run(vp, dx, dt, wav, x_s, x_r, P, **kw):
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    prop = deepwave.scalar.Propagator({"vp": vp}, dx, pml_width=P, **kw)
    return prop(wav.to(vp.device), x_s.to(vp.device), x_r.to(vp.device), dt)


def test_deepwave_cpml_mode_agrees_inside_the_model_and_reflects_less_than_the_sponge():
    """`Propagator(..., absorbing="cpml-staggered")` advances the same scalar equation as the first-order pressure-velocity
    system (the P-SV kernels in a fluid) with a convolutional PML.  (1) Without boundaries in reach the two modes are
    two discretisations of one equation: traces agree to the discretisation error.  (2) At equal width (10 cells) the
    C-PML's boundary reflection - the difference to a run on a domain so large that nothing returns - is several
    times smaller than the sponge's."""
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    dx, dt, nt, P, c, f0 = 10.0, 0.001, 900, 10, 2000.0, 15.0
    wav = deepwave.wavelets.ricker(f0, nt, dt, 1.0 / f0).reshape(-1, 1, 1)
    ang = torch.arange(8, dtype=torch.float32) * (2 * np.pi / 8)
    ring = torch.stack([200.0 * torch.cos(ang), 200.0 * torch.sin(ang)], dim=-1)[None]      # [1, 8, 2]

    def pair(n, **kw):
        vp = torch.full((n, n), c, device=DEV)
        mid = torch.tensor([[[(n // 2) * dx, (n // 2) * dx]]])
        with torch.no_grad():
            return _deepwave_run(vp, dx, dt, wav, mid, ring + mid, P, **kw).cpu().numpy()

    small_s, big_s = pair(121, absorbing="sponge"), pair(421, absorbing="sponge")
    small_c, big_c = pair(121, absorbing="cpml-staggered", pml_freq=f0), pair(421, absorbing="cpml-staggered", pml_freq=f0)
    assert np.abs(big_s).max() > 0
    # (1) same equation, two discretisations (2nd-order 5-point Laplacian vs staggered first-order system)
    assert rel_l2(big_c, big_s) <= 0.03
    # (2) what the boundary sends back (edge 600 m from the source: returns after 0.5 s)
    late = slice(520, nt)
    refl_s = np.sqrt(np.mean((small_s - big_s)[late] ** 2))
    refl_c = np.sqrt(np.mean((small_c - big_c)[late] ** 2))
    peak = np.abs(big_s).max()
    assert refl_s > 1e-4 * peak                       # the sponge's reflection is there to be beaten
    print("boundary reflection / peak: sponge %.3e, C-PML %.3e" % (refl_s / peak, refl_c / peak))
    assert refl_c <= 0.25 * refl_s, (refl_c, refl_s, peak)
    assert np.sqrt(np.mean((small_c - big_c)[:480] ** 2)) <= 1e-4 * peak      # nothing before the edge is reached


@pytest.mark.parametrize("mode", ["cpml", "cpml-staggered"])
def test_deepwave_cpml_mode_gradient_is_the_derivative_of_its_own_forward_map(mode):
    """vp.grad through the C-PML modes - "cpml": the second-order C-PML inside the scalar scheme (edge-replicated pad,
    r = vp^2 dt^2 / h^2, exact transposed adjoint of the layer's recursion); "cpml-staggered": staggered materials,
    source scaling by vp^2 at the source cell, exact transposed adjoint of the P-SV kernels - against a central
    difference of the loss along a smooth direction."""
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(5)
    nz, nx, dx, dt, nt, P = 60, 90, 10.0, 0.001, 400, 10
    base = 2000.0 + 400.0 * gaussian_filter(rng.standard_normal((nz, nx)), 6.0) * 6.0
    dv = gaussian_filter(rng.standard_normal((nz, nx)), 5.0) * 5.0 * 20.0
    wav = deepwave.wavelets.ricker(12.0, nt, dt, 1.0 / 12.0).reshape(-1, 1, 1).repeat(1, 2, 1)
    x_s = torch.tensor([[[50.0, 200.0]], [[50.0, 700.0]]])
    x_r = torch.zeros(2, 30, 2)
    x_r[..., 0] = 80.0
    x_r[..., 1] = (torch.arange(30).float() * 30.0)[None, :]

    def loss_of(v, grad):
        vp = torch.tensor(v.astype(np.float32), device=DEV, requires_grad=grad)
        rec = _deepwave_run(vp, dx, dt, wav, x_s, x_r, P, absorbing=mode, pml_freq=12.0)
        loss = 0.5 * (rec.double() ** 2).sum()
        if grad:
            loss.backward()
            return float(loss.detach()), vp.grad.double().cpu().numpy()
        return float(loss.detach()), None

    l0, g = loss_of(base, True)
    assert l0 > 0 and np.abs(g).max() > 0
    eps = 0.5
    lp, _ = loss_of(base + eps * dv, False)
    lm, _ = loss_of(base - eps * dv, False)
    fd = (lp - lm) / (2 * eps)
    an = float((g * dv).sum())
    assert abs(fd - an) <= 0.03 * abs(an), (fd, an)


def test_deepwave_pml_width_is_a_pml_with_absorbing_cpml():
    """`Propagator(..., pml_width=W, absorbing="cpml")`: the second-order C-PML inside the scalar scheme.  Inside the
    model it IS the sponge mode's scheme (same traces until anything has come back from an edge, to round-off); what
    the edge returns, measured as in the staggered test above, is 1e-3 of the direct wave's peak or less at 20 cells
    and far below the sponge's at 10 and 20 (recorded in DESIGN.md section 3)."""
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    dx, dt, nt, c, f0 = 10.0, 0.001, 900, 2000.0, 15.0
    wav = deepwave.wavelets.ricker(f0, nt, dt, 1.0 / f0).reshape(-1, 1, 1)
    ang = torch.arange(8, dtype=torch.float32) * (2 * np.pi / 8)
    ring = torch.stack([200.0 * torch.cos(ang), 200.0 * torch.sin(ang)], dim=-1)[None]

    def pair(n, P, **kw):
        vp = torch.full((n, n), c, device=DEV)
        mid = torch.tensor([[[(n // 2) * dx, (n // 2) * dx]]])
        with torch.no_grad():
            return _deepwave_run(vp, dx, dt, wav, mid, ring + mid, P, **kw).cpu().numpy()

    big = pair(421, 20)
    peak = np.abs(big).max()
    late = slice(520, nt)
    out = {}
    for P in (10, 20):
        for mode in ("sponge", "cpml"):
            small = pair(121, P, absorbing=mode, pml_freq=f0)
            assert np.sqrt(np.mean((small - big)[:480] ** 2)) <= 1e-5 * peak          # the same scheme inside the model
            out[(mode, P)] = np.abs((small - big)[late]).max() / peak
    print("edge return / peak:", {k: "%.2e" % v for k, v in out.items()})
    assert out[("cpml", 20)] <= 1e-3 and out[("cpml", 10)] <= 0.2 * out[("sponge", 10)]
    assert out[("cpml", 20)] <= 0.1 * out[("sponge", 20)]


def test_deepwave_cfl_option_reproduces_deepwaves_substep_ratio():
    """cfl="deepwave": dt_max = 0.6 / (vp_max sqrt(sum 1/dx^2)) - at dx = 10 m, dt = 1 ms the internal step halves above
    4243 m/s, where the default (0.9 x the stability limit) still takes whole steps up to
    4.9 km/s.  Both are stable and agree to the time-discretisation error; the option changes what is computed, so
    it is asserted through the traces."""
    import physicsbasedfwi2_amd.compat.deepwave as deepwave
    dx, dt, nt = 10.0, 0.001, 300
    wav = deepwave.wavelets.ricker(10.0, nt, dt, 0.1).reshape(-1, 1, 1)
    vp = torch.full((80, 80), 4500.0, device=DEV)
    x_s = torch.tensor([[[400.0, 400.0]]])
    x_r = torch.tensor([[[400.0, 600.0], [200.0, 300.0]]])
    with torch.no_grad():
        a = _deepwave_run(vp, dx, dt, wav, x_s, x_r, 10)
        b = _deepwave_run(vp, dx, dt, wav, x_s, x_r, 10, cfl="deepwave")
        c2 = _deepwave_run(vp, dx, dt, wav, x_s, x_r, 10, cfl=0.45)          # also two sub-steps
    assert a.shape == b.shape == (nt, 1, 2) and float(a.abs().max()) > 0
    e_ab = rel_l2(b.cpu().numpy(), a.cpu().numpy())
    assert 1e-4 < e_ab < 0.1                       # another inner step: different numbers, same wave
    assert torch.equal(b, c2)                      # ratio 2 either way
    with pytest.raises(Exception):
        deepwave.scalar.Propagator({"vp": vp}, dx, cfl="fastest")


def _wavesolver_setup(shape=(60, 50), spacing=(15.0, 15.0), nbpml=10, tn=600.0):
    """The set-up of acoustic_example.py:28-55 / gradient_example.py:22-54, re-typed: two-layer model, Ricker at
    10 Hz in the centre two cells below the top, receivers across x at the same depth."""
    from physicsbasedfwi2_amd.compat import seisgan_wavesolver as ws
    vp = np.full(shape, 1.5, dtype=np.float32)
    vp[:, shape[1] // 2:] = 2.5
    model = ws.Model(origin=(0.0, 0.0), spacing=spacing, shape=shape, m=1.0 / vp ** 2, nbpml=nbpml)
    dt = model.critical_dt
    nt = int(1 + tn / dt)
    time = np.linspace(0.0, tn, nt)
    src = ws.RickerSource(name="src", grid=model.grid, f0=0.01, time=time)
    src.coordinates.data[0, :] = np.array(model.domain_size) * 0.5
    src.coordinates.data[0, -1] = model.origin[-1] + 2 * spacing[-1]
    rec = ws.Receiver(name="rec", grid=model.grid, ntime=nt, npoint=shape[0])
    rec.coordinates.data[:, 0] = np.linspace(0.0, model.domain_size[0], num=shape[0])
    rec.coordinates.data[:, 1:] = src.coordinates.data[0, 1:]
    solver = ws.AcousticWaveSolver(model, source=src, receiver=rec, kernel="OT2", space_order=4, device="cuda:0")
    return ws, model, src, rec, solver, nt


def test_wavesolver_shim_runs_the_gradient_example_loop(oracle32):
    """Row a12: the four AcousticWaveSolver operations behind the reference's call signatures.  The loop of
    gradient_example.py:97-146 re-typed against the shim (its own step sizes and acceptance criterion), the forward
    data against the oracle composed by hand, adjoint and Born against the transposes they must be."""
    from scipy.ndimage import gaussian_filter
    ws, model, src, rec, solver, nt = _wavesolver_setup()
    rec_t, _, _ = solver.forward(m=model.m)                                   # true data
    d_true = rec_t.data.copy()
    # --- forward data = oracle with Devito's time loop composed by hand ------------------------------------------------
    nb, h = model.nbpml, model.spacing
    N0, N1 = model.shape_domain
    dt = model.critical_dt
    d0, d1 = H.damp_profile_1d(N0, nb, h[0]), H.damp_profile_1d(N1, nb, h[1])
    _, q0, q1, c0, c1 = H.acoustic_coeffs(np.ones((N0, N1)), d0, d1, dt, h)
    r_true = np.float32(dt * dt / (15.0 * 15.0)) / model.m.data
    f = np.zeros((nt, 1, 1), dtype=np.float32)
    f[:nt - 2, 0, 0] = src.data[1:nt - 1, 0] * np.float32(225.0)
    sc, sw = H.bilinear_taps(src.coordinates.data[None].astype(np.float64), h, nb, (N0, N1))
    rc, rw = H.bilinear_taps(rec.coordinates.data[None].astype(np.float64), h, nb, (N0, N1))
    ro = oracle32.acoustic_forward(r_true, q0.astype(np.float32), q1.astype(np.float32), f, sc, sw, rc, rw, c0, c1)
    syn_o = np.zeros((nt, rec.npoint), dtype=np.float32)
    syn_o[1:nt - 1] = ro[0:nt - 2, 0]
    assert np.abs(syn_o).max() > 0 and rel_l2(d_true, syn_o) < 1e-5
    # --- gradient_example.py: smooth start, gradient, Taylor remainders ----------------------------------------------------
    # start model = the smoothed true one with 5 % more square slowness: the residual then holds the direct wave's
    # travel-time error as well (F0 ~ 2 % of the data energy) - with the smoothing alone F0 is 1e-11 of it and its
    # changes drown in fp32 round-off.  The same set-up on the fp32 oracle gives slopes 0.946 / 1.963.
    m_true = model.m.data.copy()
    m0 = (1.05 * gaussian_filter(m_true, sigma=4.0)).astype(np.float32)
    dm = (m_true - m0).astype(np.float32)
    rec_s, u0, _ = solver.forward(save=True, m=m0)
    d0_ = rec_s.data.copy()
    residual = ws.Receiver(name="rec", grid=model.grid, ntime=nt, coordinates=rec.coordinates.data)
    residual.data[:] = d0_ - d_true
    grad = ws.Function(name="grad", grid=model.grid)
    solver.gradient(residual, u0, m=m0, grad=grad)
    with pytest.raises(Exception):
        solver.gradient(residual, u0, m=m0)                                   # the planes were consumed
    F0 = 0.5 * np.linalg.norm((d0_ - d_true).astype(np.float64)) ** 2
    G = float(np.dot(grad.data.reshape(-1).astype(np.float64), dm.reshape(-1).astype(np.float64)))
    Hs = [0.5, 0.25, .125, 0.0625, 0.0312, 0.015625, 0.0078125]                           # gradient_example.py:122
    e1, e2 = [], []
    for hh in Hs:
        d, _, _ = solver.forward(m=m0 + hh * dm)
        Fh = 0.5 * np.linalg.norm((d.data - d_true).astype(np.float64)) ** 2
        e1.append(abs(Fh - F0)); e2.append(abs(Fh - F0 - hh * G))
    p1 = np.polyfit(np.log10(Hs), np.log10(e1), 1)[0]
    p2 = np.polyfit(np.log10(Hs), np.log10(e2), 1)[0]
    assert np.isclose(p1, 1.0, rtol=0.1) and np.isclose(p2, 2.0, rtol=0.1), (p1, p2)     # gradient_example.py:143-146
    # a second shot accumulates into the same symbol (layers.py:169-183)
    g1 = grad.data.copy()
    _, u1, _ = solver.forward(save=True, m=m0)
    solver.gradient(residual, u1, m=m0, grad=grad)
    assert rel_l2(grad.data, 2.0 * g1) < 1e-6
    # --- adjoint: <F src, y> = <src, F^T y> -------------------------------------------------------------------------------
    rng = np.random.default_rng(3)
    y = ws.Receiver(name="rec", grid=model.grid, ntime=nt, coordinates=rec.coordinates.data)
    y.data[:] = rng.standard_normal(y.data.shape)
    srca, _, _ = solver.adjoint(y, m=m0)
    lhs = float(np.sum(d0_.astype(np.float64) * y.data))
    rhs = float(np.sum(srca.data.astype(np.float64) * src.data))
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs))
    # --- Born: J dm = d/de F(m0 + e dm), and <J dm, y> = <dm, J^T y> with J^T y = gradient(y) ------------------------------
    drec, _, _, _ = solver.born(dm, m=m0)
    eps = 1e-2
    dp, _, _ = solver.forward(m=m0 + eps * dm)
    dmn, _, _ = solver.forward(m=m0 - eps * dm)
    fd = (dp.data - dmn.data) / (2 * eps)
    assert rel_l2(drec.data, fd) < 2e-2
    _, u2, _ = solver.forward(save=True, m=m0)
    gy, _ = solver.gradient(y, u2, m=m0)
    lhs = float(np.sum(drec.data.astype(np.float64) * y.data))
    rhs = float(np.sum(gy.data.astype(np.float64) * dm))
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs))
    # --- what the shim does not serve fails loudly --------------------------------------------------------------------------
    with pytest.raises(Exception):
        ws.AcousticWaveSolver(model, source=src, receiver=rec, kernel="OT4", space_order=4)
    with pytest.raises(Exception):
        u0.data


def test_deepwave_shim_caches_the_cells_of_location_tensors_it_has_seen():
    """The coordinates -> cells conversion of the deepwave-shaped shim is kept per location tensor (object + version,
    device tensors only): the reference passes the same x_s / x_r to a new Propagator every iteration."""
    from physicsbasedfwi2_amd.compat.deepwave import scalar
    dev = torch.device("cuda:0")
    loc = torch.tensor([[[30.0, 50.0]], [[30.0, 70.0]]], device=dev)
    a = scalar._cells(loc, [10.0, 10.0], 4, 40, dev)
    b = scalar._cells(loc, [10.0, 10.0], 4, 40, dev)
    assert a[0] is b[0] and a[0].tolist() == [[[7 * 40 + 9]], [[7 * 40 + 11]]]
    assert scalar._cells(loc, [10.0, 10.0], 5, 40, dev)[0] is not a[0]            # another pad: another entry
    loc[0, 0, 1] = 90.0                                                              # in place: converted again
    c = scalar._cells(loc, [10.0, 10.0], 4, 40, dev)
    assert c[0] is not a[0] and c[0].tolist() == [[[7 * 40 + 13]], [[7 * 40 + 11]]]
    host = loc.cpu()
    assert scalar._cells(host, [10.0, 10.0], 4, 40, dev)[0] is not scalar._cells(host, [10.0, 10.0], 4, 40, dev)[0]
