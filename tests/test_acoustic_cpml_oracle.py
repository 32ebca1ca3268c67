"""oracle/acoustic_cpml.c (the scalar scheme with a second-order C-PML, what `deepwave.scalar.Propagator(pml_width=W)`
is for the reference: models/networks.py:5408-5411) - pinned by properties, fp64, CPU:
the layer switched off is bit for bit oracle/acoustic.c; the adjoint is the exact transpose (dot-product identity,
second-order Taylor remainder); what comes back from the edge is measured against the sponge at 10 and 20 cells."""
import numpy as np
import pytest

from oracle import helpers as H


def _profiles(n, w, h, dt, vmax, f0):
    """a, b rows (integer nodes) of the C-PML tables: [2, n]."""
    t = H.cpml_profiles(n, w, h, dt, vmax, f0)
    return np.stack([t[0], t[1]])


def _setup(n=60, w=10, nt=260, ns=2, seed=0, h=10.0, f0=0.02):
    rng = np.random.default_rng(seed)
    vp = 1.5 + 1.0 * rng.random((n, n))
    vp = H.pad_edge(vp, w)
    N = n + 2 * w
    dt = H.critical_dt((h, h), vp.max())
    r = (vp * dt / h) ** 2
    ab = _profiles(N, w, h, dt, vp.max(), f0)
    t = np.arange(nt) * dt
    f = np.zeros((nt, ns, 1))
    f[:, :, 0] = (H.ricker_seisgan(f0, t) * 100.0)[:, None] * (1.0 + 0.2 * np.arange(ns))[None]
    sz = rng.integers(w + 2, N - w - 2, (ns, 1))
    sx = rng.integers(w + 2, N - w - 2, (ns, 1))
    sc, sw = H.cell_taps(sz, sx, N)
    rz = np.full((ns, 9), w + 3)
    rx = np.linspace(2, N - 3, 9).astype(int)[None, :].repeat(ns, 0)      # the outer ones sit inside the layer
    rc, rw = H.cell_taps(rz, rx, N)
    return dict(r=r, ab=ab, f=f, sc=sc, sw=sw, rc=rc, rw=rw, dt=dt, N=N)


def test_layer_switched_off_is_the_undamped_scheme_bit_for_bit(oracle64, oracle32):
    for o in (oracle64, oracle32):
        c = _setup(nt=120)
        off = np.zeros_like(c["ab"])
        a, Ga = o.acoustic_cpml_forward(c["r"], off, off, c["f"], c["sc"], c["sw"], c["rc"], c["rw"], save=True)
        z = np.zeros(c["N"])
        b, Gb = o.acoustic_forward(c["r"], z, z, c["f"], c["sc"], c["sw"], c["rc"], c["rw"], save=True)
        assert np.abs(a).max() > 0 and np.array_equal(a, b) and np.array_equal(Ga, Gb)
        g = np.sign(a)
        ga, fa = o.acoustic_cpml_backward(c["r"], off, off, c["sc"], c["sw"], c["rc"], c["rw"], g, Ga)
        gb, fb = o.acoustic_backward(c["r"], z, z, c["sc"], c["sw"], c["rc"], c["rw"], g, Gb)
        assert np.array_equal(ga, gb) and np.array_equal(fa, fb)


def test_adjoint_is_the_exact_transpose(oracle64):
    o = oracle64
    c = _setup(seed=3)
    geo = (c["sc"], c["sw"], c["rc"], c["rw"])
    rng = np.random.default_rng(5)
    # source -> seismogram map (linear): <J q, d> = <q, J^T d>
    q = rng.standard_normal(c["f"].shape)
    rec, G = o.acoustic_cpml_forward(c["r"], c["ab"], c["ab"], q, *geo, save=True)
    d = rng.standard_normal(rec.shape)
    _, gq = o.acoustic_cpml_backward(c["r"], c["ab"], c["ab"], *geo, d, G)
    lhs, rhs = np.sum(rec * d), np.sum(q * gq)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    # model gradient: second-order Taylor remainder of 0.5 ||rec - obs||^2 in (r, f)
    rec, G = o.acoustic_cpml_forward(c["r"], c["ab"], c["ab"], c["f"], *geo, save=True)
    obs = rec + 0.3 * np.abs(rec).max() * rng.standard_normal(rec.shape)
    gr, gf = o.acoustic_cpml_backward(c["r"], c["ab"], c["ab"], *geo, rec - obs, G)
    dr = c["r"] * 0.02 * rng.standard_normal(c["r"].shape)
    df = 0.05 * np.abs(c["f"]).max() * rng.standard_normal(c["f"].shape)
    J0 = 0.5 * np.sum((rec - obs) ** 2)
    lin = np.sum(gr * dr) + np.sum(gf * df)
    e1, e2, hs = [], [], [1e-2, 1e-3, 1e-4]
    for hh in hs:
        a = o.acoustic_cpml_forward(c["r"] + hh * dr, c["ab"], c["ab"], c["f"] + hh * df, *geo)
        Jh = 0.5 * np.sum((a - obs) ** 2)
        e1.append(abs(Jh - J0)); e2.append(abs(Jh - J0 - hh * lin))
    p1 = np.polyfit(np.log10(hs), np.log10(e1), 1)[0]
    p2 = np.polyfit(np.log10(hs), np.log10(e2), 1)[0]
    assert abs(p1 - 1.0) < 0.05 and abs(p2 - 2.0) < 0.05, (p1, p2)


def _edge_return(o, kind, w, n=160, h=10.0, f0=0.015, vp0=2.0, nt=520):
    """Peak of what the absorbing layer sends back, relative to the peak of the direct wave, at a receiver 25 cells
    from the layer and 30 cells to the side of a source 15 cells from it (homogeneous model; the reference run has
    the same interior inside a 120-cell frame that nothing returns from in time)."""
    def run(width, absorber, frame):
        N = n + 2 * frame
        dt = H.critical_dt((h, h), vp0)
        r = np.full((N, N), (vp0 * dt / h) ** 2)
        t = np.arange(nt) * dt
        f = (H.ricker_seisgan(f0, t) * 100.0)[:, None, None]
        sc, sw = H.cell_taps([[frame + 15]], [[frame + n // 2]], N)
        rc, rw = H.cell_taps([[frame + 25]], [[frame + n // 2 + 30]], N)
        if absorber == "cpml":
            ab = _profiles(N, width, h, dt, vp0, f0)
            return o.acoustic_cpml_forward(r, ab, ab, f, sc, sw, rc, rw)[:, 0, 0]
        d = H.damp_profile_1d(N, width, h)
        _, q0, q1, _, _ = H.acoustic_coeffs(np.ones((N, N)), d, d, dt, (h, h))
        return o.acoustic_forward(r, q0, q1, f, sc, sw, rc, rw)[:, 0, 0]
    ref = run(40, "sponge", 120)
    got = run(w, kind, w)
    return np.abs(got - ref).max() / np.abs(ref).max()


def test_what_comes_back_from_the_edge(oracle64):
    """Measured (15 Hz Ricker, 13 cells per wavelength, source 15 cells from the layer):
       sponge 10 cells 1.05e-1, 20 cells 4.3e-2;   C-PML 10 cells 1.1e-3, 20 cells 1.6e-4
    - the 20-cell C-PML meets the 1e-3 asked of it, the sponge of model.py:6-29 needs far more than 20 cells to."""
    got = {(k, w): _edge_return(oracle64, k, w) for k in ("sponge", "cpml") for w in (10, 20)}
    print(got)
    assert got[("cpml", 20)] < 1e-3 and got[("cpml", 10)] < 1e-2
    assert got[("cpml", 10)] < 0.05 * got[("sponge", 10)] and got[("cpml", 20)] < 0.05 * got[("sponge", 20)]
    assert np.isfinite(list(got.values())).all()
