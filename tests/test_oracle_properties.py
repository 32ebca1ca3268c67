"""Pin the CPU oracle with the known-answer / property tests the reference holds for this path:
  * analytical 2-D Green's function and the published space-order-4 error table
    (seisgan/fwi/pde/seismic/acoustic/accuracy.ipynb cells 9-12, 18);
  * Taylor gradient test, slopes 1 and 2 within rtol 0.1 (gradient_example.py:115-146);
plus build-authored adjoint dot-product, reciprocity, absorbing-layer and fp32/fp64 checks.
"""
import numpy as np
import pytest

from cases import acoustic_case, elastic_case, rel_l2
from oracle import helpers as H


# ---------------------------------------------------------------------------------------------
# accuracy.ipynb: c0 = 1.5 km/s, f0 = 0.07 kHz, dt = 0.1 ms, 1501 samples, source (200,200) m,
# receiver (260,260) m, nbpml = 40, source scaled by 100/(c0 h)^2.  Devito's time loop runs
# time = 1..nt-2 with src[time] injected into u[time+1] and rec[time] read from u[time]
# (SURVEY.md appendix C); the mapping onto the oracle's loop is  f[n] = src[n+1],
# rec_devito[t] = rec[t-1].
def _notebook_run(o, nn, h, nt=1501, dt=0.1, c0=1.5, f0=0.07, nb=40):
    N = nn + 2 * nb
    m = np.full((N, N), 1.0 / c0 ** 2)
    d = H.damp_profile_1d(N, nb, h)
    r, q0, q1, k0, k1 = H.acoustic_coeffs(m, d, d, dt, (h, h))
    t = np.arange(nt) * dt
    rr = np.pi * f0 * (t - 1.0 / f0)                    # upstream Ricker of the notebook (1/f0)
    src = 100.0 * (1 - 2 * rr ** 2) * np.exp(-rr ** 2) / (c0 * h) ** 2
    f = np.zeros((nt, 1, 1))
    f[:nt - 2, 0, 0] = src[1:nt - 1] * h * h
    sc, sw = H.bilinear_taps(np.array([[[200.0, 200.0]]]), (h, h), nb, (N, N))
    rc, rw = H.bilinear_taps(np.array([[[260.0, 260.0]]]), (h, h), nb, (N, N))
    rec = o.acoustic_forward(r, q0, q1, f, sc, sw, rc, rw, k0, k1)
    out = np.zeros(nt)
    out[1:nt - 1] = rec[0:nt - 2, 0, 0]
    return out


def test_analytical_green_function_and_published_error_table(oracle64):
    U = H.analytical_2d(0.07, 1.5, np.hypot(60.0, 60.0), 30001, 0.1, 1.0 / 0.07, amp=1e2)[:1501]
    published = {2.0: 0.035965613536, 2.5: 0.0906846693164, 4.0: 0.533946654328}
    err = {}
    for nn, h in ((201, 2.0), (161, 2.5), (101, 4.0)):
        err[h] = np.abs(_notebook_run(oracle64, nn, h) - U).max()
        # the notebook measures against its own order-20 run, whose L-inf distance to the analytical solution
        # is 7.83e-3 (reproduced in test_reference_pins.py, where the table itself is matched to 0.2 %): by
        # the triangle inequality the error against the analytical solution is within that of the table
        assert abs(err[h] - published[h]) <= 8e-3, (h, err[h])
    order = np.log(err[4.0] / err[2.0]) / np.log(2.0)
    assert 3.6 <= order <= 4.4            # notebook: observed order ~3.9
    assert err[2.0] / np.abs(U).max() < 0.02


# ---------------------------------------------------------------------------------------------
def _taylor(J, grad_dot, hs):
    e1, e2 = [], []
    J0 = J(0.0)
    for h in hs:
        Jh = J(h)
        e1.append(abs(Jh - J0))
        e2.append(abs(Jh - J0 - h * grad_dot))
    p1 = np.polyfit(np.log10(hs), np.log10(e1), 1)[0]
    p2 = np.polyfit(np.log10(hs), np.log10(e2), 1)[0]
    return p1, p2


def test_acoustic_taylor_slopes(oracle64):
    """gradient_example.py:115-146: data from the 'true' model, m0 = smooth model, dm = true - m0;
    error1 = |Phi(m0+h dm) - Phi(m0)| has slope 1, error2 = |... - h <grad, dm>| slope 2."""
    o = oracle64
    c = acoustic_case(seed=2, ntap=4, nt=110)
    args = (c["q0"], c["q1"])
    geo = (c["sc"], c["sw"], c["rc"], c["rw"], c["c0"], c["c1"])
    rng = np.random.default_rng(0)
    from scipy.ndimage import gaussian_filter
    dr = c["r"] * 0.08 * gaussian_filter(rng.standard_normal(c["r"].shape), 3.0) * 3.0
    obs = o.acoustic_forward(c["r"] + dr, *args, c["f"], *geo)
    rec, G = o.acoustic_forward(c["r"], *args, c["f"], *geo, save=True)
    gr, _ = o.acoustic_backward(c["r"], *args, c["sc"], c["sw"], c["rc"], c["rw"], rec - obs, G,
                                c["c0"], c["c1"], want_grad_f=False)

    def J(h):
        rr = o.acoustic_forward(c["r"] + h * dr, *args, c["f"], *geo)
        return 0.5 * np.sum((rr - obs) ** 2)
    # gradient_example.py:121 uses H = 0.5 .. 0.0078; with data generated from m0 + dm the
    # Gauss-Newton term makes error1 = h A (1 - h/2), so the same 7 halvings start at 0.125
    H_ = [0.125, 0.0625, 0.0312, 0.015625, 0.0078125, 0.00390625, 0.001953125]
    p1, p2 = _taylor(J, np.sum(gr * dr), H_)
    assert np.isclose(p1, 1.0, rtol=0.1) and np.isclose(p2, 2.0, rtol=0.1)
    # exact discrete adjoint (model and source amplitudes): clean second-order remainder
    df = rng.standard_normal(c["f"].shape) * np.abs(c["f"]).max() * 0.05
    gr, gf = o.acoustic_backward(c["r"], *args, c["sc"], c["sw"], c["rc"], c["rw"], rec - obs, G,
                                 c["c0"], c["c1"])

    def J2(h):
        rr = o.acoustic_forward(c["r"] + h * dr, *args, c["f"] + h * df, *geo)
        return 0.5 * np.sum((rr - obs) ** 2)
    _, p2 = _taylor(J2, np.sum(gr * dr) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02


def test_acoustic_adjoint_dot_product(oracle64):
    """<F q, d> = <q, F^T d> for the map source amplitudes -> receiver traces."""
    o = oracle64
    c = acoustic_case(seed=6, nsrc=2, ntap=4)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"], c["c0"], c["c1"])
    rng = np.random.default_rng(1)
    q = rng.standard_normal(c["f"].shape)
    rec, G = o.acoustic_forward(c["r"], c["q0"], c["q1"], q, *geo, save=True)
    d = rng.standard_normal(rec.shape)
    _, gq = o.acoustic_backward(c["r"], c["q0"], c["q1"], c["sc"], c["sw"], c["rc"], c["rw"], d, G,
                                c["c0"], c["c1"])
    lhs, rhs = np.sum(rec * d), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_acoustic_born_is_the_derivative_and_the_transpose_of_the_gradient(oracle64):
    """Born operator (operators.py:168-207): J dr equals the central finite difference of the forward
    map to O(eps^2), and <J dr, g> = <dr, J^T g> with J^T the gradient operator (operators.py:127-165)."""
    o = oracle64
    c = acoustic_case(seed=9, nsrc=1, ntap=4, nt=120)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"], c["c0"], c["c1"])
    rng = np.random.default_rng(2)
    dr = rng.standard_normal(c["r"].shape) * c["r"] * 0.05
    rec, G = o.acoustic_forward(c["r"], c["q0"], c["q1"], c["f"], *geo, save=True)
    jdr = o.acoustic_born(c["r"], c["q0"], c["q1"], dr, G, c["rc"], c["rw"], c["c0"], c["c1"])
    errs = []
    for eps in (1e-2, 1e-3):
        fp = o.acoustic_forward(c["r"] + eps * dr, c["q0"], c["q1"], c["f"], *geo)
        fm = o.acoustic_forward(c["r"] - eps * dr, c["q0"], c["q1"], c["f"], *geo)
        errs.append(rel_l2((fp - fm) / (2 * eps), jdr))
    assert np.abs(jdr).max() > 0 and errs[1] < 1e-6 and errs[0] / errs[1] > 50      # second order in eps
    g = rng.standard_normal(rec.shape)
    gr, _ = o.acoustic_backward(c["r"], c["q0"], c["q1"], c["sc"], c["sw"], c["rc"], c["rw"], g, G,
                                c["c0"], c["c1"])
    lhs, rhs = np.sum(jdr * g), np.sum(dr * gr)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_acoustic_reciprocity(oracle64):
    o = oracle64
    c = acoustic_case(seed=8, ns=1, nrec=1, nt=160)
    f = c["f"]
    a = o.acoustic_forward(c["r"], c["q0"], c["q1"], f, c["sc"], c["sw"], c["rc"], c["rw"])
    # swap source and receiver: with the source term scaled by r[cell] = vp^2 dt^2/h^2 the
    # discrete operator is symmetric, so the two traces coincide
    b = o.acoustic_forward(c["r"], c["q0"], c["q1"], f, c["rc"], c["rw"], c["sc"], c["sw"])
    assert np.abs(a).max() > 0 and rel_l2(a, b) < 1e-10


def test_sponge_absorbs(oracle64):
    o = oracle64
    c = acoustic_case(seed=3, n0=60, n1=60, nb=20, nt=1400, ns=1, nrec=5, f0=0.03)
    rec = o.acoustic_forward(c["r"], c["q0"], c["q1"], c["f"], c["sc"], c["sw"], c["rc"], c["rw"])
    assert np.abs(rec[-100:]).max() < 0.02 * np.abs(rec).max()


def test_acoustic_fp32_tracks_fp64(oracle32, oracle64):
    c = acoustic_case(seed=4)
    a = [o.acoustic_forward(c["r"], c["q0"], c["q1"], c["f"], c["sc"], c["sw"], c["rc"], c["rw"])
         for o in (oracle32, oracle64)]
    assert rel_l2(a[0], a[1]) < 1e-5


# ---------------------------------------------------------------------------------------------
def test_elastic_taylor_slopes(oracle64):
    o = oracle64
    c = elastic_case(seed=1)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rng = np.random.default_rng(0)
    from scipy.ndimage import gaussian_filter
    dm = c["mat"] * 0.05 * gaussian_filter(rng.standard_normal(c["mat"].shape), (0, 3, 3)) * 3.0
    ox, oz = o.elastic_forward(c["mat"] + dm, c["pz"], c["px"], c["f"], *geo)
    vx, vz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], *geo, save=True)
    gm, gf = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, vx - ox, vz - oz, S)

    def J(h):
        a, b = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], c["f"], *geo)
        return 0.5 * np.sum((a - ox) ** 2) + 0.5 * np.sum((b - oz) ** 2)
    p1, p2 = _taylor(J, np.sum(gm * dm), [0.125, 0.0625, 0.0312, 0.015625, 0.0078125, 0.00390625,
                                          0.001953125])
    assert np.isclose(p1, 1.0, rtol=0.1) and np.isclose(p2, 2.0, rtol=0.1)
    df = rng.standard_normal(c["f"].shape) * np.abs(c["f"]).max() * 0.05

    def J2(h):
        a, b = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], c["f"] + h * df, *geo)
        return 0.5 * np.sum((a - ox) ** 2) + 0.5 * np.sum((b - oz) ** 2)
    _, p2 = _taylor(J2, np.sum(gm * dm) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02


def test_elastic_adjoint_dot_product(oracle64):
    o = oracle64
    c = elastic_case(seed=5, nsrc=2)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rng = np.random.default_rng(2)
    q = rng.standard_normal(c["f"].shape)
    vx, vz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], q, *geo, save=True)
    dx, dz = rng.standard_normal(vx.shape), rng.standard_normal(vz.shape)
    _, gq = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, dx, dz, S)
    lhs, rhs = np.sum(vx * dx) + np.sum(vz * dz), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_cpml_absorbs_and_is_stable(oracle64):
    o = oracle64
    c = elastic_case(seed=7, nz=70, nx=70, fw=10, nt=1500, ns=1, nrec=5, water=0, freq=10.0)
    vx, vz = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], c["sc"], c["sw"], c["rc"], c["rw"])
    assert np.isfinite(vx).all()
    assert np.abs(vz[-150:]).max() < 0.03 * np.abs(vz).max()
    # without the layer the energy stays in the box
    nop_z, nop_x = H.cpml_profiles(70, 0, 20.0, 0.002, 3000.0, 5.0), H.cpml_profiles(70, 0, 20.0, 0.002, 3000.0, 5.0)
    wx, wz = o.elastic_forward(c["mat"], nop_z, nop_x, c["f"], c["sc"], c["sw"], c["rc"], c["rw"])
    assert np.abs(wz[-150:]).max() > 0.2 * np.abs(wz).max()


def test_elastic_homogeneous_pressure_wave_speed(oracle64):
    """Explosive source in a homogeneous solid: first arrival at offset d travels with vp."""
    o = oracle64
    nz = nx = 120
    h, dt, vp0 = 10.0, 0.001, 3000.0
    mat = H.elastic_materials(np.full((nz, nx), vp0), np.full((nz, nx), vp0 / np.sqrt(3)),
                              np.full((nz, nx), 2000.0), dt, h)
    pz, px = H.cpml_profiles(nz, 10, h, dt, vp0, 10.0), H.cpml_profiles(nx, 10, h, dt, vp0, 10.0)
    nt = 300
    f = (H.ricker_deepwave(25.0, nt, dt, 0.05) * 1e6)[:, None, None]
    sc, sw = H.cell_taps([[60]], [[30]], nx)
    rc, rw = H.cell_taps([[60]], [[90]], nx)
    vx, _ = o.elastic_forward(mat, pz, px, f, sc, sw, rc, rw)
    t_peak = np.argmax(np.abs(vx[:, 0, 0])) * dt
    expect = 0.05 + 60 * h / vp0
    assert abs(t_peak - expect) < 0.012


@pytest.mark.parametrize("water", [0, 6])
def test_elastic_free_surface_adjoint_is_exact(oracle64, water):
    """FREE_SURF=1 (networks.py:9811): odd stress mirroring + szz(0)=0; the transposed scheme must
    give a clean second-order Taylor remainder and pass the dot-product test."""
    o = oracle64
    c = elastic_case(seed=11, free_surface=True, water=water, nsrc=2)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rng = np.random.default_rng(3)
    vx, vz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], *geo, save=True, free_surface=1)
    ox = vx + rng.standard_normal(vx.shape) * 0.3 * np.abs(vx).max()
    oz = vz + rng.standard_normal(vz.shape) * 0.3 * np.abs(vz).max()
    gm, gf = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, vx - ox, vz - oz, S, free_surface=1)
    dm = rng.standard_normal(c["mat"].shape) * c["mat"] * 0.02
    df = rng.standard_normal(c["f"].shape) * np.abs(c["f"]).max() * 0.05

    def J(h):
        a, b = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], c["f"] + h * df, *geo,
                                 free_surface=1)
        return 0.5 * np.sum((a - ox) ** 2) + 0.5 * np.sum((b - oz) ** 2)
    _, p2 = _taylor(J, np.sum(gm * dm) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02
    q = rng.standard_normal(c["f"].shape)
    a, b, S2 = o.elastic_forward(c["mat"], c["pz"], c["px"], q, *geo, save=True, free_surface=1)
    dx, dz = rng.standard_normal(a.shape), rng.standard_normal(b.shape)
    _, gq = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, dx, dz, S2, free_surface=1)
    lhs, rhs = np.sum(a * dx) + np.sum(b * dz), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_free_surface_reflects(oracle64):
    """With the free surface the top reflects (energy stays longer) while C-PML on top absorbs."""
    o = oracle64
    kw = dict(seed=7, nz=70, nx=70, fw=10, nt=700, ns=1, nrec=5, water=0, freq=10.0)
    cf, ca = elastic_case(free_surface=True, **kw), elastic_case(free_surface=False, **kw)
    for c in (cf, ca):          # same source a few rows below the top, same receivers
        c["sc"], c["sw"] = H.cell_taps([[6]], [[35]], 70)
        c["rc"], c["rw"] = H.cell_taps([[30] * 5], [[15, 25, 35, 45, 55]], 70)
    vf = o.elastic_forward(cf["mat"], cf["pz"], cf["px"], cf["f"], cf["sc"], cf["sw"], cf["rc"], cf["rw"],
                           free_surface=1)[1]
    va = o.elastic_forward(ca["mat"], ca["pz"], ca["px"], ca["f"], ca["sc"], ca["sw"], ca["rc"], ca["rw"])[1]
    assert np.isfinite(vf).all()
    assert np.sum(vf ** 2) > 1.2 * np.sum(va ** 2)


@pytest.mark.parametrize("source_type,free_surface", [(1, 0), (2, 0), (2, 1)])
def test_elastic_force_sources_adjoint_is_exact(oracle64, source_type, free_surface):
    """Point forces (DENISE QUELLTYPB 2 / 3, networks.py:10419-10453 uses 2): the amplitude enters vx / vz
    between V and S; dot-product identity for the source -> seismogram map and a second-order Taylor
    remainder of the objective in (materials, source)."""
    o = oracle64
    c = elastic_case(seed=13, free_surface=bool(free_surface), nsrc=2)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    kw = dict(free_surface=free_surface, source_type=source_type)
    rng = np.random.default_rng(4)
    f = c["f"] * 1e-3                                  # a force moves the medium far more than a moment rate
    vx, vz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], f, *geo, save=True, **kw)
    assert np.abs(vx).max() > 0 and np.abs(vz).max() > 0
    ex, ez = o.elastic_forward(c["mat"], c["pz"], c["px"], f, *geo, free_surface=free_surface)
    assert not np.allclose(vx, ex)                     # not the explosive response
    ox = vx + rng.standard_normal(vx.shape) * 0.3 * np.abs(vx).max()
    oz = vz + rng.standard_normal(vz.shape) * 0.3 * np.abs(vz).max()
    gm, gf = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, vx - ox, vz - oz, S, **kw)
    dm = rng.standard_normal(c["mat"].shape) * c["mat"] * 0.02
    df = rng.standard_normal(f.shape) * np.abs(f).max() * 0.05

    def J(h):
        a, b = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], f + h * df, *geo, **kw)
        return 0.5 * np.sum((a - ox) ** 2) + 0.5 * np.sum((b - oz) ** 2)
    _, p2 = _taylor(J, np.sum(gm * dm) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02
    q = rng.standard_normal(f.shape)
    a, b, S2 = o.elastic_forward(c["mat"], c["pz"], c["px"], q, *geo, save=True, **kw)
    dx, dz = rng.standard_normal(a.shape), rng.standard_normal(b.shape)
    _, gq = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, dx, dz, S2, **kw)
    lhs, rhs = np.sum(a * dx) + np.sum(b * dz), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_elastic_pressure_receivers_adjoint_is_exact(oracle64):
    """Pressure seismograms (DENISE SEISMO 2 / 4): sum w (sxx + szz) after the stress update; dot-product
    identity of the source -> (vx, vz, p) map and a second-order Taylor remainder with p in the objective."""
    o = oracle64
    c = elastic_case(seed=17, nsrc=2, free_surface=True)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rng = np.random.default_rng(6)
    vx, vz, S, p = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], *geo, save=True, free_surface=1,
                                     pressure=True)
    assert np.abs(p).max() > 0
    op = p + rng.standard_normal(p.shape) * 0.3 * np.abs(p).max()
    gm, gf = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, np.zeros_like(vx), np.zeros_like(vz), S,
                                free_surface=1, g_p=p - op)
    dm = rng.standard_normal(c["mat"].shape) * c["mat"] * 0.02
    df = rng.standard_normal(c["f"].shape) * np.abs(c["f"]).max() * 0.05

    def J(h):
        q = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], c["f"] + h * df, *geo, free_surface=1,
                              pressure=True)[2]
        return 0.5 * np.sum((q - op) ** 2)
    _, p2 = _taylor(J, np.sum(gm * dm) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02
    q = rng.standard_normal(c["f"].shape)
    a, b, S2, pp = o.elastic_forward(c["mat"], c["pz"], c["px"], q, *geo, save=True, free_surface=1, pressure=True)
    dx, dz, dp = (rng.standard_normal(a.shape) for _ in range(3))
    _, gq = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, dx, dz, S2, free_surface=1, g_p=dp)
    lhs, rhs = np.sum(a * dx) + np.sum(b * dz) + np.sum(pp * dp), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_elastic_second_order_stencils_adjoint_is_exact(oracle64):
    """fd_order = 2 (DENISE FD_ORDER, left commented at networks.py:10447): weights (1, 0) in the same four-point
    form.  The transposed scheme passes the dot-product identity and leaves a second-order Taylor remainder; the
    seismograms differ from the fourth-order ones (more numerical dispersion), both stay bounded."""
    o = oracle64
    c = elastic_case(seed=19, nsrc=2, free_surface=True)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    kw = dict(free_surface=1, fd_order=2)
    rng = np.random.default_rng(8)
    vx, vz, S = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], *geo, save=True, **kw)
    vx4, vz4 = o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"], *geo, free_surface=1)
    assert np.isfinite(vx).all() and np.abs(vx).max() > 0
    assert 1e-3 < rel_l2(vx, vx4) < 1.0
    ox = vx + rng.standard_normal(vx.shape) * 0.3 * np.abs(vx).max()
    oz = vz + rng.standard_normal(vz.shape) * 0.3 * np.abs(vz).max()
    gm, gf = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, vx - ox, vz - oz, S, **kw)
    dm = rng.standard_normal(c["mat"].shape) * c["mat"] * 0.02
    df = rng.standard_normal(c["f"].shape) * np.abs(c["f"]).max() * 0.05

    def J(h):
        a, b = o.elastic_forward(c["mat"] + h * dm, c["pz"], c["px"], c["f"] + h * df, *geo, **kw)
        return 0.5 * np.sum((a - ox) ** 2) + 0.5 * np.sum((b - oz) ** 2)
    _, p2 = _taylor(J, np.sum(gm * dm) + np.sum(gf * df), [1e-2, 1e-3, 1e-4])
    assert abs(p2 - 2.0) < 0.02
    q = rng.standard_normal(c["f"].shape)
    a, b, S2 = o.elastic_forward(c["mat"], c["pz"], c["px"], q, *geo, save=True, **kw)
    dx, dz = rng.standard_normal(a.shape), rng.standard_normal(b.shape)
    _, gq = o.elastic_backward(c["mat"], c["pz"], c["px"], *geo, dx, dz, S2, **kw)
    lhs, rhs = np.sum(a * dx) + np.sum(b * dz), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    # back to the default weights for whoever runs next in this process
    o.elastic_forward(c["mat"], c["pz"], c["px"], c["f"][:2], *geo, free_surface=1)


def test_elastic_stencil_orders_converge_as_advertised(oracle64):
    """Explosive source in a homogeneous solid, pressure receiver on a grid node, the same physical set-up on
    h = 20, 10 and 5 m with one small time step (time error out of the picture): against the fine fourth-order
    run the second-order stencil loses a factor ~4 of error per halving of h, the fourth-order one ~16."""
    o = oracle64
    vp0, rho0, f0, dt = 3000.0, 2000.0, 12.0, 2.5e-4
    nt = 960

    def run(n, order):
        h = 1200.0 / n
        mat = H.elastic_materials(np.full((n, n), vp0), np.full((n, n), vp0 / np.sqrt(3)), np.full((n, n), rho0), dt, h)
        nop = H.cpml_profiles(n, 0, h, dt, vp0, 10.0)
        t = np.arange(nt) * dt
        a = (np.pi * f0 * (t - 0.1)) ** 2
        f = ((1 - 2 * a) * np.exp(-a) * dt / h ** 2 * 1e6)[:, None, None]       # moment rate density
        sc, sw = H.cell_taps([[n // 2]], [[n // 3]], n)
        rc, rw = H.cell_taps([[n // 2]], [[n // 3 + n // 4]], n)
        return o.elastic_forward(mat, nop, nop, f, sc, sw, rc, rw, fd_order=order, pressure=True)[2][:, 0, 0]
    ref = run(240, 4)
    err = {(order, n): np.linalg.norm(run(n, order) - ref) / np.linalg.norm(ref) for order in (2, 4) for n in (60, 120)}
    r2, r4 = err[(2, 60)] / err[(2, 120)], err[(4, 60)] / err[(4, 120)]
    assert 3.0 < r2 < 5.5 and r4 > 10.0 and err[(4, 120)] < 0.1 * err[(2, 120)], err
