"""Seeded synthetic cases shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
import numpy as np

from oracle import helpers as H


def acoustic_case(seed=0, n0=40, n1=50, nb=8, nt=90, ns=2, nsrc=1, nrec=7, ntap=1,
                  h=(10.0, 10.0), f0=0.02, dt_scale=1.0):
    """Random smooth-ish model in seisgan units (m, ms, km/s); returns a dict of numpy arrays
    in the parametrisation of oracle/acoustic.c."""
    rng = np.random.default_rng(seed)
    vp = 1.5 + 1.5 * rng.random((n0, n1))
    m = H.pad_edge(1.0 / vp ** 2, nb)
    N0, N1 = m.shape
    s = H.critical_dt(h, vp.max()) * dt_scale
    d0 = H.damp_profile_1d(N0, nb, h[0])
    d1 = H.damp_profile_1d(N1, nb, h[1])
    r, q0, q1, c0, c1 = H.acoustic_coeffs(m, d0, d1, s, h)
    t = np.arange(nt) * s
    f = np.zeros((nt, ns, nsrc))
    for i in range(nsrc):
        f[:, :, i] = (H.ricker_seisgan(f0 * (1 + 0.3 * i), t) * 100.0)[:, None]
    f *= (1.0 + 0.1 * np.arange(ns))[None, :, None]
    ext0, ext1 = (n0 - 1) * h[0], (n1 - 1) * h[1]
    src_xy = np.zeros((ns, nsrc, 2))
    src_xy[..., 0] = rng.uniform(0.1 * ext0, 0.9 * ext0, (ns, nsrc))
    src_xy[..., 1] = rng.uniform(0.1 * ext1, 0.9 * ext1, (ns, nsrc))
    rec_xy = np.zeros((ns, nrec, 2))
    rec_xy[..., 0] = np.linspace(0.03 * ext0, 0.97 * ext0, nrec)[None, :]
    rec_xy[..., 1] = rng.uniform(0.0, ext1, (ns, 1))
    if ntap == 4:
        sc, sw = H.bilinear_taps(src_xy, h, nb, (N0, N1))
        rc, rw = H.bilinear_taps(rec_xy, h, nb, (N0, N1))
    else:
        sc, sw = H.cell_taps(np.floor(src_xy[..., 0] / h[0]).astype(int) + nb,
                             np.floor(src_xy[..., 1] / h[1]).astype(int) + nb, N1)
        rc, rw = H.cell_taps(np.floor(rec_xy[..., 0] / h[0]).astype(int) + nb,
                             np.floor(rec_xy[..., 1] / h[1]).astype(int) + nb, N1)
    return dict(r=r, q0=q0, q1=q1, c0=c0, c1=c1, f=f, sc=sc, sw=sw, rc=rc, rw=rw, s=s,
                shape=(N0, N1), nb=nb, h=h, vp=vp)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0)


def elastic_case(seed=0, nz=44, nx=60, fw=8, nt=120, ns=2, nsrc=1, nrec=9, h=20.0, dt=0.002,
                 water=6, freq=8.0, free_surface=False):
    """Random elastic model with a water layer, sources inside the top C-PML."""
    rng = np.random.default_rng(seed)
    vp = 1800 + 1500 * rng.random((nz, nx))
    vs = vp / np.sqrt(3) * (0.8 + 0.4 * rng.random((nz, nx)))
    rho = 1800 + 600 * rng.random((nz, nx))
    if water:
        vs[:water] = 0.0
        vp[:water] = 1500.0
        rho[:water] = 1000.0
    mat = H.elastic_materials(vp, vs, rho, dt, h, free_surface=free_surface)
    pz = H.cpml_profiles(nz, fw, h, dt, 3000.0, 5.0, lo=not free_surface)
    px = H.cpml_profiles(nx, fw, h, dt, 3000.0, 5.0)
    f = np.zeros((nt, ns, nsrc))
    for i in range(nsrc):
        f[:, :, i] = (H.ricker_deepwave(freq * (1 + 0.2 * i), nt, dt, 1.2 / freq) * 1e6)[:, None]
    f *= (1.0 + 0.1 * np.arange(ns))[None, :, None]
    sz = rng.integers(0 if free_surface else 2, 5, (ns, nsrc))
    if free_surface:
        sz[0, 0] = 0           # a source ON the free surface: szz(0,.) must stay 0
    sx = rng.integers(4, nx - 4, (ns, nsrc))
    sc, sw = H.cell_taps(sz, sx, nx)
    rz_ = np.full((ns, nrec), min(nz - 3, water + 14))
    rx_ = np.linspace(2, nx - 3, nrec).astype(int)[None, :].repeat(ns, 0)
    rc, rw = H.cell_taps(rz_, rx_, nx)
    if free_surface:           # a receiver spread on the surface row exercises the mirrored rows
        rc[0], rw[0] = H.cell_taps(np.zeros((1, nrec), dtype=int), rx_[:1], nx)
    return dict(mat=mat, pz=pz, px=px, f=f, sc=sc, sw=sw, rc=rc, rw=rw, fw=fw, vp=vp, vs=vs,
                rho=rho, dt=dt, h=h, fs=1 if free_surface else 0)
