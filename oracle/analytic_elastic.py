"""ORACLE -- TEST INFRASTRUCTURE ONLY: closed-form 2-D P-SV solutions that pin oracle/elastic.c.

Nothing here restates reference code: DENISE is absent from /root/reference, so the elastic
scheme is pinned to the textbook solutions of the equations it discretises instead:

  rho v_t = div sigma + F(t) delta(x) delta(z - zs) e_k                  (line force, `force_x` / `force_z`)
  sigma_t = lambda (div v) I + 2 mu eps(v) + s(t) delta(x) delta(z - zs) I   (explosive line source, `explosive`)

in a homogeneous full space or under a stress-free surface z = 0 (z down), by the Cagniard - de Hoop method
(de Hoop 1960; Aki & Richards, Quantitative Seismology, ch. 6; Garvin 1956 for the buried explosive line source,
Lamb 1904 for the line force).  With u = grad phi + curl (psi e_y) and every field written as
(s / 2 pi i) int (.) exp(-s p x) dp, a source radiates the potentials

  explosive:  A+- = S^ / (2 rho s eta_a alpha^2),   B+- = 0
  force_z:    A+- = -+ F^ / (2 rho s^2),            B+- = -F^ p / (2 rho s^2 eta_b)
  force_x:    A+- = -F^ p / (2 rho s^2 eta_a),      B+- = +- F^ / (2 rho s^2)           (+: below the source)

(eta_c = sqrt(1/c^2 - p^2)), the free surface returns an up-going (a, b) as

  A = [(X - g^2) a + 4 beta^2 p eta_b g b] / R,     B = [-4 beta^2 p eta_a g a + (X - g^2) b] / R,
  g = 1 - 2 beta^2 p^2,  X = 4 beta^4 p^2 eta_a eta_b,  R = g^2 + X   (Rayleigh function),

and each of the six generalised rays (P, S direct; PP, PS, SP, SS reflected) gives, for a source time function
whose derivative is W,  v(T) = (1/pi) Im int_path W(T - t(p)) K(p) dp  along its Cagniard path t(p) = p x + sum eta d
real.  The integral is taken in p (parametrised by tau, t = t_saddle + tau^2), where the integrand is smooth: the
inverse-square-root wavefront singularities of the time-domain Green's function never appear.  Rays whose saddle
lies beyond 1/alpha get the real-axis segment [1/alpha, p_saddle] as well (head waves).

`fullspace_hankel` is an independent frequency-domain evaluation (Kupradze's tensor with Hankel functions) of the
full-space cases; `garvin_surface` is Garvin's closed form for a receiver ON the surface.  The tests require all
three to agree before anything is compared with the finite-difference oracle.
"""
import numpy as np


def _eta(p, c):
    """sqrt(1/c^2 - p^2) on the sheet Re >= 0, continued from the first quadrant of p (-i sqrt(..) for real p > 1/c)."""
    e = np.sqrt(1.0 / (c * c) - np.asarray(p, dtype=complex) ** 2)
    return np.where(e.imag > 0, np.conj(e), e)


class _Medium:
    def __init__(self, alpha, beta, rho):
        self.a, self.b, self.rho = float(alpha), float(beta), float(rho)

    def parts(self, p):
        ea, eb = _eta(p, self.a), _eta(p, self.b)
        g = 1.0 - 2.0 * self.b ** 2 * p * p
        X = 4.0 * self.b ** 4 * p * p * ea * eb
        return ea, eb, g, X, g * g + X


def _source_potentials(med, kind, p, below):
    """(A, B) without the transform of the time function and its powers of s (see the module header)."""
    ea, eb = _eta(p, med.a), _eta(p, med.b)
    sg = 1.0 if below else -1.0
    if kind == "explosive":
        return 1.0 / (2.0 * med.rho * ea * med.a ** 2), 0.0 * p
    if kind == "force_z":
        return -sg / (2.0 * med.rho) + 0.0 * p, -p / (2.0 * med.rho * eb)
    if kind == "force_x":
        return -p / (2.0 * med.rho * ea), sg / (2.0 * med.rho) + 0.0 * p
    raise ValueError(kind)


def _rays(med, kind, x, zs, zr, free_surface):
    """List of (legs, K) - legs = [(c, d), ...] of the phase t(p) = p x + sum eta_c d, K(p) -> (Kx, Kz)."""
    a, b = med.a, med.b
    down = zr > zs
    eps = 1.0 if down else -1.0
    d0 = abs(zr - zs)

    def direct_p(p):
        A, _ = _source_potentials(med, kind, p, down)
        return -p * A, -eps * _eta(p, a) * A

    def direct_s(p):
        _, B = _source_potentials(med, kind, p, down)
        return eps * _eta(p, b) * B, -p * B
    rays = [([(a, d0)], direct_p)]
    if kind != "explosive":
        rays.append(([(b, d0)], direct_s))
    if not free_surface:
        return rays

    def refl(p):
        ea, eb, g, X, R = med.parts(p)
        A, B = _source_potentials(med, kind, p, False)          # the up-going pair
        return ea, eb, (X - g * g) / R, 4 * b * b * p * eb * g / R, -4 * b * b * p * ea * g / R, A, B

    def pp(p):
        ea, eb, rpp, rsp, rps, A, B = refl(p)
        return -p * rpp * A, -ea * rpp * A

    def sp(p):
        ea, eb, rpp, rsp, rps, A, B = refl(p)
        return -p * rsp * B, -ea * rsp * B

    def ps(p):
        ea, eb, rpp, rsp, rps, A, B = refl(p)
        return eb * rps * A, -p * rps * A

    def ss(p):
        ea, eb, rpp, rsp, rps, A, B = refl(p)
        return eb * rpp * B, -p * rpp * B
    rays.append(([(a, zs), (a, zr)], pp))
    rays.append(([(a, zs), (b, zr)], ps))
    if kind != "explosive":
        rays.append(([(b, zs), (a, zr)], sp))
        rays.append(([(b, zs), (b, zr)], ss))
    return rays


def _phase(p, x, legs):
    t = p * x
    for c, d in legs:
        if d > 0:
            t = t + _eta(p, c) * d
    return t


def _dphase(p, x, legs):
    t = x + 0.0 * p
    for c, d in legs:
        if d > 0:
            t = t - p * d / _eta(p, c)
    return t


def _saddle(x, legs):
    live = [(c, d) for c, d in legs if d > 0]
    if not live:
        raise ValueError("source and receiver at the same depth on this ray: degenerate Cagniard path")
    pmax = min(1.0 / c for c, _ in live)
    if len(set(c for c, _ in live)) == 1:
        D = sum(d for _, d in live)
        return x / (np.hypot(x, D) * live[0][0])
    lo, hi = 0.0, pmax
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if _dphase(mid, x, live).real > 0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def _path(x, legs, t_end, n):
    """Points of the complex part of the Cagniard path: tau in [0, sqrt(t_end - t0)] (Gauss-Legendre panels),
    p(tau) with t(p) = t0 + tau^2, dp/dtau, weights."""
    live = [(c, d) for c, d in legs if d > 0]
    p0 = _saddle(x, legs)
    t0 = _phase(p0, x, live).real
    if t_end <= t0:
        return p0, t0, None
    # second derivative at the saddle (negative): t'' = -sum d / (c^2 eta^3)
    t2 = -sum(d / (c * c * _eta(p0, c).real ** 3) for c, d in live)
    tau_max = np.sqrt(t_end - t0)
    npan = max(8, n // 8)
    gx, gw = np.polynomial.legendre.leggauss(8)
    edges = np.linspace(0.0, tau_max, npan + 1)
    tau = (0.5 * (edges[:-1] + edges[1:])[:, None] + 0.5 * np.diff(edges)[:, None] * gx[None, :]).ravel()
    w = (0.5 * np.diff(edges)[:, None] * gw[None, :]).ravel()
    p = np.empty(tau.size, dtype=complex)
    prev_tau, prev_p = 0.0, complex(p0)
    slope = 1j * np.sqrt(2.0 / abs(t2))                         # dp/dtau at the saddle
    for k, tk in enumerate(tau):
        q = prev_p + slope * (tk - prev_tau)
        target = t0 + tk * tk
        for _ in range(50):
            F = _phase(q, x, live) - target
            dF = _dphase(q, x, live)
            step = F / dF
            q = q - step
            if q.imag < 0:                                      # stay on the first-quadrant branch
                q = complex(q.real, abs(q.imag))
            if abs(step) < 1e-15 * (abs(q) + 1e-300):
                break
        p[k] = q
        if tk > prev_tau:
            slope = (q - prev_p) / (tk - prev_tau)
        prev_tau, prev_p = tk, q
    dpdtau = 2.0 * tau / _dphase(p, x, live)
    return p0, t0, (tau, p, dpdtau, w)


def velocity(kind, alpha, beta, rho, x, zs, zr, times, wavelet_rate, free_surface=True, n=1600):
    """v_x, v_z at (x, zr) and the given times for a unit line source of `kind` at (0, zs): `wavelet_rate(t)` is the
    time derivative of F(t) (forces) or of s(t) (explosive source), vectorised.  x may be negative."""
    med = _Medium(alpha, beta, rho)
    flip = x < 0
    x = abs(float(x))
    if x == 0:
        raise ValueError("x = 0: use a small offset (the path parametrisation assumes x > 0)")
    times = np.asarray(times, dtype=float)
    vx, vz = np.zeros(times.size), np.zeros(times.size)
    t_end = times.max()
    for legs, K in _rays(med, kind, x, float(zs), float(zr), free_surface):
        p0, t0, path = _path(x, legs, t_end, n)
        if path is not None:
            tau, p, dpdtau, w = path
            kx, kz = K(p)
            Wm = wavelet_rate(times[:, None] - (t0 + tau * tau)[None, :])
            vx += (Wm * (kx * dpdtau * w).imag[None, :]).sum(1) / np.pi
            vz += (Wm * (kz * dpdtau * w).imag[None, :]).sum(1) / np.pi
        if p0 > 1.0 / med.a:                                    # real-axis segment: head waves
            live = [(c, d) for c, d in legs if d > 0]
            gx, gw = np.polynomial.legendre.leggauss(400)
            qmax = np.sqrt(p0 - 1.0 / med.a)
            q = 0.5 * qmax * (gx + 1.0)
            pr = 1.0 / med.a + q * q
            wr = 0.5 * qmax * gw * 2.0 * q
            kx, kz = K(pr)
            tr = _phase(pr, x, live).real
            Wm = wavelet_rate(times[:, None] - tr[None, :])
            vx += (Wm * (np.asarray(kx).imag * wr)[None, :]).sum(1) / np.pi
            vz += (Wm * (np.asarray(kz).imag * wr)[None, :]).sum(1) / np.pi
    if flip:                                                    # mirror x -> -x
        if kind == "force_x":
            vz = -vz
        else:
            vx = -vx
    return vx, vz


def garvin_surface(alpha, beta, rho, x, zs, times, wavelet_rate, n=1600):
    """Garvin's problem in closed form: explosive line source at depth zs, receiver ON the free surface.  One
    Cagniard path (a single P leg): p(t) = [x t + i zs sqrt(t^2 - r^2/alpha^2)] / r^2, and
    u_z^ = gamma / (rho alpha^2 R),  u_x^ = -2 p beta^2 eta_b / (rho alpha^2 R)."""
    med = _Medium(alpha, beta, rho)
    times = np.asarray(times, dtype=float)
    r = np.hypot(x, zs)
    t0 = r / alpha
    th_max = np.arccosh(max(times.max() / t0, 1.0 + 1e-12))
    gx, gw = np.polynomial.legendre.leggauss(n)
    th = 0.5 * th_max * (gx + 1.0)
    w = 0.5 * th_max * gw
    p = t0 * (x * np.cosh(th) + 1j * zs * np.sinh(th)) / r ** 2
    dp = t0 * (x * np.sinh(th) + 1j * zs * np.cosh(th)) / r ** 2
    ea, eb, g, X, R = med.parts(p)
    kz = g / (rho * alpha ** 2 * R)
    kx = -2.0 * p * beta ** 2 * eb / (rho * alpha ** 2 * R)
    Wm = wavelet_rate(times[:, None] - (t0 * np.cosh(th))[None, :])
    vx = (Wm * (kx * dp * w).imag[None, :]).sum(1) / np.pi
    vz = (Wm * (kz * dp * w).imag[None, :]).sum(1) / np.pi
    return vx, vz


def fullspace_hankel(kind, alpha, beta, rho, x, z, times, wavelet, pad=8):
    """Independent evaluation of the full-space cases in the frequency domain.  `wavelet(t)` is F(t) or s(t) itself.
    Force:  G_ij = [k_b^2 g_b delta_ij + d_i d_j (g_b - g_a)] / (rho w^2),  g_c = -(i/4) H0^(2)(k_c r)  (Kupradze);
    explosive:  v = grad(phi_t),  phi^ = S^ g_a / (rho alpha^2)."""
    from scipy.special import hankel2
    times = np.asarray(times, dtype=float)
    dt = times[1] - times[0]
    nfft = int(2 ** np.ceil(np.log2(times.size * pad)))
    t = times[0] + dt * np.arange(nfft)
    W = np.fft.rfft(wavelet(t))
    om = 2 * np.pi * np.fft.rfftfreq(nfft, dt)
    om[0] = 1.0
    r = np.hypot(x, z)
    nx_, nz_ = x / r, z / r

    def g(c):
        return -0.25j * hankel2(0, om / c * r)

    def dg(c):          # d/dr
        return 0.25j * (om / c) * hankel2(1, om / c * r)

    def d2g(c):         # d2/dr2 = -k^2 g - g'/r
        return -(om / c) ** 2 * g(c) - dg(c) / r
    if kind == "explosive":
        # v = d/dt grad phi, phi^ = S^ g_a/(rho alpha^2), S^ = s^/(i w):  v^ = s^ grad g_a / (rho alpha^2)
        vr = W * dg(alpha) / (rho * alpha ** 2)
        ux, uz = vr * nx_, vr * nz_
    else:
        j = (nx_, nz_)[0 if kind == "force_x" else 1]
        dd = d2g(beta) - d2g(alpha)
        d1 = (dg(beta) - dg(alpha)) / r
        out = []
        for i, ni in enumerate((nx_, nz_)):
            dij = 1.0 if i == (0 if kind == "force_x" else 1) else 0.0
            hess = ni * j * dd + (dij - ni * j) * d1
            G = ((om / beta) ** 2 * g(beta) * dij + hess) / (rho * om ** 2)
            out.append(1j * om * G * W)                         # velocity
        ux, uz = out
    ux[0] = uz[0] = 0.0
    return np.fft.irfft(ux, nfft)[:times.size], np.fft.irfft(uz, nfft)[:times.size]
