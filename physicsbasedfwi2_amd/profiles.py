"""Host-side scalar/1-D set-up of a survey: absorbing-layer profiles, wavelets, stability
limits and coordinate -> cell conventions.  Pure numpy/torch on tiny arrays (init only);
the per-cell, per-step arithmetic lives in the HIP library.

Reference behaviour mirrored here (paths relative to the reference tree):
  damping layer      seisgan/fwi/pde/seismic/model.py:6-29
  critical dt        seisgan/fwi/pde/seismic/model.py:160-168
  Ricker (seisgan)   seisgan/fwi/pde/seismic/source.py:224-231   (peak at 2/f0)
  Ricker (deepwave)  deepwave.wavelets.ricker as called at models/networks.py:5357
  TimeAxis           seisgan/fwi/pde/seismic/source.py:18-69
"""
import math

import numpy as np
import torch


# ------------------------------------------------------------------------------ damping --
def sponge_profile(n, width, h):
    """One axis of the seisgan damping field: layer cell i (0 = outermost) carries
    1.5 ln(1000)/40 * (p - sin(2 pi p)/(2 pi)) / h with p = (width - i + 1)/width, mirrored
    at the far end; the 2-D field is the sum of the two axes (corners add)."""
    prof = np.zeros(n, dtype=np.float64)
    if width <= 0:
        return prof
    k = np.arange(width, dtype=np.float64)
    p = np.abs((width - k + 1.0) / float(width))
    val = (1.5 * math.log(1.0 / 0.001) / 40.0) * (p - np.sin(2.0 * np.pi * p) / (2.0 * np.pi)) / h
    prof[:width] += val
    prof[n - width:] += val[::-1]
    return prof


def sponge_q(n, width, h_axis, h_ref, dt):
    """q = damp * h_ref^2 / (2 dt): the dimensionless form the stencil kernel consumes."""
    return sponge_profile(n, width, h_axis) * (h_ref * h_ref) / (2.0 * dt)


# ----------------------------------------------------------------------------- stability --
def seisgan_critical_dt(spacing, vp_max):
    """0.42 * min(h) / max(vp) in 2-D (0.38 in 3-D)."""
    coeff = 0.38 if len(spacing) == 3 else 0.42
    return coeff * min(spacing) / vp_max


def scalar_cfl_limit(spacing, vp_max):
    """Largest stable dt of the 2nd-order-time / 4th-order-space scalar scheme:
    dt <= 2 / (vp sqrt(sum_k 16/(3 h_k^2)))   (sum |D2 weights| = 16/3 per axis)."""
    s = sum(16.0 / (3.0 * h * h) for h in spacing)
    return 2.0 / (vp_max * math.sqrt(s))


def elastic_cfl_limit(h, vp_max, fd_order=4):
    """Staggered grid in 2-D: dt <= h / (vp sqrt(2) sum|c_k|) - (9/8 + 1/24) for order 4, 1 for order 2."""
    w = 1.0 if int(fd_order) == 2 else 9.0 / 8.0 + 1.0 / 24.0
    return h / (vp_max * math.sqrt(2.0) * w)


# ------------------------------------------------------------------------------ time axis --
def time_axis(start, stop, step):
    """seisgan TimeAxis with `num` derived: num = ceil((stop-start+step)/step); the stop value
    is then re-derived from num (it may exceed the requested stop)."""
    num = int(math.ceil((stop - start + step) / step))
    return num, step * (num - 1) + start


def time_axis_complete(start=None, step=None, num=None, stop=None):
    """The four quantities of a uniform time axis from any three of them (seisgan TimeAxis, source.py:36-58):
    stop = start + step (num - 1).  With `num` missing it is derived as :func:`time_axis` derives it (and the stop
    value follows from it); given all four, or fewer than three, there is nothing to solve for."""
    missing = [k for k, v in (("start", start), ("step", step), ("num", num), ("stop", stop)) if v is None]
    if len(missing) != 1:
        raise ValueError("exactly three of start, step, num and stop must be given (missing: %s)"
                         % (", ".join(missing) or "none"))
    if missing[0] == "num":
        num, stop = time_axis(start, stop, step)
    elif missing[0] == "start":
        start = stop - step * (num - 1)
    elif missing[0] == "step":
        step = (stop - start) / (num - 1)
    else:
        stop = start + step * (num - 1)
    if not isinstance(num, int):
        raise TypeError("num must be an int, got %s" % type(num).__name__)
    return start, step, num, stop


# -------------------------------------------------------------------------------- wavelets --
def ricker_seisgan(f0, t):
    """(1 - 2 r^2) exp(-r^2), r = pi f0 (t - 2/f0); t and 1/f0 in the same unit."""
    t = np.asarray(t, dtype=np.float64)
    r = math.pi * f0 * (t - 2.0 / f0)
    return (1.0 - 2.0 * r * r) * np.exp(-r * r)


def ricker(freq, length, dt, peak_time, dtype=torch.float32):
    """deepwave.wavelets.ricker signature: torch tensor [length]."""
    t = torch.arange(int(length), dtype=torch.float64) * dt - peak_time
    a = (math.pi * freq * t) ** 2
    return ((1.0 - 2.0 * a) * torch.exp(-a)).to(dtype)


# ------------------------------------------------------------------ coordinates -> cells --
def cells_truncate(loc, spacing, pad, n1):
    """deepwave convention: cell = trunc(loc / dx) per dimension, dimension order = model
    tensor order; `pad` cells of absorbing layer precede the physical origin.
    loc [..., 2] (float) -> int32 linear cells [..., 1] and unit weights."""
    loc = torch.as_tensor(loc, dtype=torch.float64)
    i0 = torch.trunc(loc[..., 0] / spacing[0]).to(torch.int64) + pad
    i1 = torch.trunc(loc[..., 1] / spacing[1]).to(torch.int64) + pad
    cells = (i0 * n1 + i1).to(torch.int32).unsqueeze(-1)
    return cells, torch.ones(cells.shape, dtype=torch.float32)


def cells_round(x, y, dh, n1, pad0=0, pad1=0):
    """DENISE convention: node = iround(coordinate / DH), 1-based, y = depth -> 0-based cell
    (iy-1, ix-1) of a [nz, nx] array."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    ix = np.floor(x / dh + 0.5).astype(np.int64) - 1 + pad1
    iy = np.floor(y / dh + 0.5).astype(np.int64) - 1 + pad0
    return iy, ix, (iy * n1 + ix).astype(np.int32)


def cells_bilinear(coords, spacing, pad, shape_pad):
    """Devito sparse operator: (bi)linear over the enclosing cell, coordinates relative to
    the un-padded origin (operators.py:81-85, offset=nbpml).  coords [..., 2] ->
    cells [..., 4] int32 (-1 = inactive), weights [..., 4]."""
    c = np.asarray(coords, dtype=np.float64)
    p0, p1 = c[..., 0] / spacing[0], c[..., 1] / spacing[1]
    i0, i1 = np.floor(p0).astype(np.int64), np.floor(p1).astype(np.int64)
    a0, a1 = p0 - i0, p1 - i1
    n0, n1 = shape_pad
    j0, j1 = i0 + pad, i1 + pad
    cells = np.stack([j0 * n1 + j1, j0 * n1 + j1 + 1, (j0 + 1) * n1 + j1,
                      (j0 + 1) * n1 + j1 + 1], axis=-1)
    w = np.stack([(1 - a0) * (1 - a1), (1 - a0) * a1, a0 * (1 - a1), a0 * a1], axis=-1)
    rows = np.stack([j0, j0, j0 + 1, j0 + 1], axis=-1)
    cols = np.stack([j1, j1 + 1, j1, j1 + 1], axis=-1)
    ok = (w != 0) & (rows >= 0) & (rows < n0) & (cols >= 0) & (cols < n1)
    cells = np.where(ok, cells, -1).astype(np.int32)
    w = np.where(ok, w, 0.0)
    return torch.from_numpy(cells), torch.from_numpy(w.astype(np.float32))


# ------------------------------------------------------------------------------- C-PML --
def cpml_tables(n, width, h, dt, vpml, fpml, npower=4.0, kmax=1.0, low=True, high=True,
                rcoef=0.0008):
    """1-D convolutional-PML tables [6, n] = a, b, 1/kappa at integer nodes, then the same
    three at half nodes (x = (i+1/2) h).  DENISE-style layer INSIDE the grid, `width` nodes
    (pyapi_denise names: FW, DAMPING = vpml, FPML, npower, k_max_PML).
      abscissa: low side (width - x/h) h, high side (x/h - (n-1-width)) h, clipped at 0
      d = d0 (abscissa/L)^N,  d0 = -(N+1) vpml ln(rcoef)/(2L),  L = width*h
      kappa = 1 + (kmax-1)(abscissa/L)^N,  alpha = pi fpml (1 - abscissa/L)
      b = exp(-(d/kappa + alpha) dt),  a = d (b-1)/(kappa (d + kappa alpha))
    Outside the layer a = b = 0 and 1/kappa = 1 (the kernels then skip the memory variable)."""
    out = np.zeros((6, n), dtype=np.float64)
    out[2] = 1.0
    out[5] = 1.0
    if width <= 0 or not (low or high):
        return out
    L = width * h
    d0 = -(npower + 1.0) * vpml * math.log(rcoef) / (2.0 * L)
    alpha_max = math.pi * fpml
    for half in (0, 1):
        pos = np.arange(n, dtype=np.float64) + 0.5 * half
        absc = np.zeros(n)
        if low:
            absc = np.maximum(absc, (width - pos) * h)
        if high:
            absc = np.maximum(absc, (pos - (n - 1 - width)) * h)
        nrm = absc / L
        d = d0 * nrm ** npower
        kappa = 1.0 + (kmax - 1.0) * nrm ** npower
        alpha = alpha_max * (1.0 - np.minimum(nrm, 1.0))
        b = np.exp(-(d / kappa + alpha) * dt)
        den = kappa * (d + kappa * alpha)
        safe = den > 1e-30
        a = np.where(safe, d * (b - 1.0) / np.where(safe, den, 1.0), 0.0)
        inside = nrm > 0
        out[3 * half + 0] = np.where(inside, a, 0.0)
        out[3 * half + 1] = np.where(inside, b, 0.0)
        out[3 * half + 2] = np.where(inside, 1.0 / kappa, 1.0)
    return out
