"""``import physicsbasedfwi2_amd.compat.pyapi_denise as api`` -- the subset of the (author-modified)
pyapi_denise protocol that models/networks.py drives (import at line 31; canonical call sequence
7603-7606, 7665-7666, 7698-7731, 7752-7761, 7787-7802; SEAM variant 9686-9877):

    d = api.Denise(root, verbose=1); d.save_folder = ...; d.set_paths(); d.help()
    d.NPROCX = 6; d.PHYSICS = 1; d.TIME = 5.0; d.FC_SPIKE_1 = ...          # plain attributes
    model = api.Model(vp, vs, rho, dx); src = api.Sources(x, y, f); rec = api.Receivers(x, y)
    d.fwi_stages = []; d.add_fwi_stage(fc_high=10, inv_rho_iter=10000)
    d.grad(model, src, rec)                      # reference: mpirun of 30 DENISE ranks + files
    loss = np.loadtxt('loss_curve_grad.out')
    grads, names = d.get_fwi_gradients(['seis'], return_filenames=True)   # [rho, vp, vs]

Here ``grad`` runs the HIP elastic propagator (forward, stage filter + L2 residual, exact adjoint,
chain rule to Vp/Vs/rho) on the current HIP device; nothing is written except
``loss_curve_grad.out`` (the caller reads it back).  Differences a maintainer must know:
  * observed data come from tensors (``d.set_observed(vx, vy)``) or from SU files through
    ``read_su`` -- DENISE reads them from ``DATA_DIR`` itself;
  * MPI decomposition attributes (NPROCX/NPROCY) are accepted and ignored: shots, not
    sub-domains, are the unit of parallelism (physicsbasedfwi2_amd.dist);
  * DENISE's binaries are not available to pin against: wavelet, C-PML constants,
    source scaling and the Butterworth stage filter follow the published formulas, and the
    gradient is the exact discrete one (the caller rescales it by max|model|/max|grad| anyway,
    networks.py:7843-7862).
"""
import math
import os
import struct

import numpy as np
import torch

from .. import elastic, misfit, profiles
from .._lib import MifwiError


class Model:
    """api.Model(vp, vs, rho, dx): 2-D arrays [ny, nx] in m/s and kg/m^3.  The caller passes them
    ``np.flipud``-ed (networks.py:7587-7589), i.e. row 0 is the DEEPEST row."""

    def __init__(self, vp, vs, rho, dx):
        self.vp = np.asarray(vp, dtype=np.float32)
        self.vs = np.asarray(vs, dtype=np.float32)
        self.rho = np.asarray(rho, dtype=np.float32)
        if not (self.vp.shape == self.vs.shape == self.rho.shape) or self.vp.ndim != 2:
            raise MifwiError("vp, vs, rho must be 2-D arrays of one shape")
        self.dx = float(dx)
        self.ny, self.nx = self.vp.shape

    def __repr__(self):
        return ("vp:\t%s, %.4f, %.4f m/s\nvs:\t%s, %.4f, %.4f m/s\nrho:\t%s, %.4f, %.4f kg/m3\n"
                "dx:\t%.4f\nSize:\n\tOX:\tmin %.4f\tmax %.4f m\n\tOZ:\tmin %.4f\tmax %.4f m"
                % (self.vp.shape, self.vp.min(), self.vp.max(), self.vs.shape, self.vs.min(),
                   self.vs.max(), self.rho.shape, self.rho.min(), self.rho.max(), self.dx, 0.0,
                   self.nx * self.dx, 0.0, self.ny * self.dx))


class _Points:
    def __init__(self, x, y):
        self.x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        self.y = np.atleast_1d(np.asarray(y, dtype=np.float64))
        if self.x.shape != self.y.shape:
            raise MifwiError("x and y must have one shape")

    def __len__(self):
        return int(self.x.size)


class Receivers(_Points):
    """api.Receivers(xrec, yrec): metres, y = depth."""


class Sources(_Points):
    """api.Sources(xsrc, ysrc, fsource): one shot per source; ``f`` = centre frequency [Hz]."""

    def __init__(self, x, y, f, td=0.0, amp=1.0, wavelets=None):
        super().__init__(x, y)
        self.f = np.broadcast_to(np.asarray(f, dtype=np.float64), self.x.shape).copy()
        self.td = np.broadcast_to(np.asarray(td, dtype=np.float64), self.x.shape).copy()
        self.amp = np.broadcast_to(np.asarray(amp, dtype=np.float64), self.x.shape).copy()
        # QUELLART = 3 (source time functions "from file"): [nshot, nt] or [nt] samples at the run's DT
        self.wavelets = None if wavelets is None else np.asarray(wavelets, dtype=np.float64)


def ricker_denise(fc, nt, dt, td=0.0):
    """SOFI2D / DENISE source time function QUELLART=1:
    tau = pi (t - 1.5/fc - td) / (1.5/fc);  s = (1 - 4 tau^2) exp(-2 tau^2)."""
    t = np.arange(nt, dtype=np.float64) * dt
    ts = 1.0 / fc
    tau = np.pi * (t - 1.5 * ts - td) / (1.5 * ts)
    return (1.0 - 4.0 * tau * tau) * np.exp(-2.0 * tau * tau)


def spike_denise(nt, dt, fc1, fc2, order=5, td=0.0):
    """QUELLART = 6: a unit spike at t = td, band-limited with the Butterworth response between
    FC_SPIKE_1 (high-pass corner, <= 0: none) and FC_SPIKE_2 (low-pass corner), order ORDER_SPIKE, causal as
    DENISE's time-domain filter (:func:`butterworth`)."""
    w = np.zeros(nt, dtype=np.float64)
    w[min(nt - 1, max(0, int(round(td / dt))))] = 1.0 / dt
    return butterworth(torch.tensor(w), dt, max(0.0, float(fc1)), float(fc2), int(order)).numpy()


def gradient_taper(ny, dh, gradt1=21, gradt2=25, gradt3=490, gradt4=500, exponent=0.0):
    """SWS_TAPER_GRAD_HOR = 1: depth window applied to every gradient, row 0 = surface.  Zero above
    grid row GRADT1, cosine ramp to one at GRADT2, one down to GRADT3, cosine ramp to zero at GRADT4
    (rows counted from 1 as in DENISE.inp; a ramp beyond the grid is simply not reached), times the
    depth preconditioner (row * DH) ** EXP_TAPER_GRAD_HOR.  DENISE's taper_grad.c is not available
    here: this is the documented stand-in (the reference mutes rows 0:25 and rescales every gradient
    to max(model)/max(grad) afterwards, networks.py:7808-7862, so only the shape matters)."""
    j = np.arange(1, ny + 1, dtype=np.float64)
    w = np.ones(ny, dtype=np.float64)
    up = (j - gradt1) / max(1.0, float(gradt2 - gradt1))
    w = np.where(j <= gradt1, 0.0, np.where(j < gradt2, 0.5 * (1.0 - np.cos(np.pi * up)), w))
    dn = (j - gradt3) / max(1.0, float(gradt4 - gradt3))
    w = np.where(j >= gradt4, 0.0, np.where(j > gradt3, w * 0.5 * (1.0 + np.cos(np.pi * dn)), w))
    if exponent:
        w = w * (j * dh) ** float(exponent)
    return w.astype(np.float32)


def butterworth(x, dt, fc_low=0.0, fc_high=0.0, order=6, zero_phase=False):
    """Butterworth filter along time (dim 0), applied in the frequency domain; differentiable (torch.fft), so the
    adjoint sources are filtered with the transposed (time-reversed) filter by autograd.
    fc_high > 0: low-pass corner, fc_low > 0: high-pass corner (both > 0: band-pass), as the FWI-stage filter of
    ``add_fwi_stage`` and the band-limited spike of QUELLART = 6.

    Default: the CAUSAL recursive filter DENISE applies in the time domain (forward only): the digital design of
    ``scipy.signal.butter`` (bilinear transform, second-order sections), evaluated as its exact complex response
    on the FFT bins of the trace padded to four times its length - equal to ``scipy.signal.sosfilt`` on the same
    samples to the decay of the impulse response.  ``zero_phase=True`` applies the analog magnitude response
    only.  DENISE's own coefficients cannot be checked here (not in the reference tree)."""
    lo = float(fc_low) if fc_low and fc_low > 0 else 0.0
    hi = float(fc_high) if fc_high and fc_high > 0 else 0.0
    if lo <= 0 and hi <= 0:
        return x
    nt = x.shape[0]
    if zero_phase:
        nfft = 2 * nt
        f = torch.fft.rfftfreq(nfft, d=dt).to(x.device)
        h = torch.ones_like(f)
        if hi > 0:
            h = h / torch.sqrt(1.0 + (f / hi) ** (2 * order))
        if lo > 0:
            h = h * torch.sqrt(1.0 / (1.0 + (lo / torch.clamp(f, min=1e-12)) ** (2 * order)))
    else:
        from scipy import signal
        fnyq = 0.5 / dt
        if hi >= fnyq:
            hi = 0.0                                   # nothing to cut below Nyquist
        if lo <= 0 and hi <= 0:
            return x
        nfft = 4 * nt
        w = np.fft.rfftfreq(nfft, d=dt) * (2.0 * np.pi * dt)          # rad / sample
        hc = np.ones(w.size, dtype=np.complex128)
        if hi > 0:
            hc = hc * signal.sosfreqz(signal.butter(order, hi / fnyq, "lowpass", output="sos"), worN=w)[1]
        if lo > 0:
            hc = hc * signal.sosfreqz(signal.butter(order, lo / fnyq, "highpass", output="sos"), worN=w)[1]
        h = torch.tensor(hc, dtype=torch.complex64 if x.dtype == torch.float32 else torch.complex128, device=x.device)
    spec = torch.fft.rfft(x, n=nfft, dim=0) * h.view(-1, *([1] * (x.dim() - 1)))
    return torch.fft.irfft(spec, n=nfft, dim=0)[:nt]


def read_su(path, endian=None, headers=False):
    """SU reader for the `.su.shot<k>` gathers DENISE writes and the reference copies around (networks.py:7669-7692):
    a sequence of traces, each a 240-byte SEG-Y trace header followed by ns IEEE float32 samples, no file header.
    Header words read (1-based byte positions of the SEG-Y / SU trace header): ns at 115-116 and dt [microseconds]
    at 117-118, both unsigned 16-bit; with ``headers=True`` also tracl 1-4, fldr 9-12, scalco 71-72, sx 73-76,
    sy 77-80, gx 81-84, gy 85-88 (signed).  ``endian``: '<' (native SU on x86, what DENISE writes), '>' (XDR SU /
    SEG-Y order) or None = whichever makes the file a whole number of equal traces (little-endian when both do).
    Returns (data [ntraces, ns] float32, dt seconds) and the header dict when asked.  A file that is not a whole
    number of traces in either byte order raises."""
    raw = open(path, "rb").read()
    if len(raw) < 240:
        raise MifwiError("%s: %d bytes, shorter than one SU trace header" % (path, len(raw)))

    def fits(e):
        ns = struct.unpack_from(e + "H", raw, 114)[0]
        return ns > 0 and len(raw) % (240 + 4 * ns) == 0

    if endian is None:
        ok = [e for e in ("<", ">") if fits(e)]
        if not ok:
            raise MifwiError("%s: not a whole number of SU traces in either byte order (ns = %d / %d, %d bytes)"
                             % (path, struct.unpack_from("<H", raw, 114)[0], struct.unpack_from(">H", raw, 114)[0],
                                len(raw)))
        endian = ok[0]
    elif endian not in ("<", ">") or not fits(endian):
        raise MifwiError("%s: not a whole number of SU traces with byte order %r" % (path, endian))
    ns = struct.unpack_from(endian + "H", raw, 114)[0]
    dt_us = struct.unpack_from(endian + "H", raw, 116)[0]
    tl = 240 + 4 * ns
    ntr = len(raw) // tl
    out = np.empty((ntr, ns), dtype=np.float32)
    hdr = {k: np.empty(ntr, dtype=np.int64) for k in ("tracl", "fldr", "scalco", "sx", "sy", "gx", "gy", "ns", "dt")}
    for i in range(ntr):
        o = i * tl
        if struct.unpack_from(endian + "H", raw, o + 114)[0] != ns:
            raise MifwiError("%s: trace %d has another sample count than trace 0" % (path, i))
        out[i] = np.frombuffer(raw, dtype=endian + "f4", count=ns, offset=o + 240)
        if headers:
            hdr["tracl"][i], hdr["fldr"][i] = struct.unpack_from(endian + "i", raw, o)[0], struct.unpack_from(endian + "i", raw, o + 8)[0]
            hdr["scalco"][i] = struct.unpack_from(endian + "h", raw, o + 70)[0]
            hdr["sx"][i], hdr["sy"][i], hdr["gx"][i], hdr["gy"][i] = struct.unpack_from(endian + "4i", raw, o + 72)
            hdr["ns"][i], hdr["dt"][i] = ns, struct.unpack_from(endian + "H", raw, o + 116)[0]
    return (out, dt_us * 1e-6, hdr) if headers else (out, dt_us * 1e-6)


def write_su(path, data, dt):
    """Inverse of :func:`read_su` (tests and hand-over of synthetic observed data)."""
    data = np.asarray(data, dtype="<f4")
    with open(path, "wb") as fh:
        for i, tr in enumerate(data):
            hdr = bytearray(240)
            struct.pack_into("<i", hdr, 0, i + 1)
            struct.pack_into("<H", hdr, 114, tr.size)
            struct.pack_into("<H", hdr, 116, int(round(dt * 1e6)))
            fh.write(bytes(hdr))
            fh.write(tr.tobytes())


def _nice_dt(limit):
    """Largest of {1, 2, 2.5, 4, 5} x 10^k below 0.9 * limit."""
    target = 0.9 * limit
    k = math.floor(math.log10(target))
    best = 0.0
    for kk in (k - 1, k):
        for m in (1.0, 2.0, 2.5, 4.0, 5.0):
            v = m * 10.0 ** kk
            if v <= target:
                best = max(best, v)
    return best


# ---- what an assignment `d.NAME = value` means here -----------------------------------------------------------------
# pyapi_denise is an attribute bag written into DENISE's parameter file; the reference configures a run by assignment only
# (networks.py:7603-7731, 9686-9833, 10419-10505, 10899-11090).  A silent bag would accept a parameter this library does
# not serve - or a typo - and return gradients of another problem, so every name falls in one of four classes:
# acted on by the kernels and the host chain around them
_SERVED = frozenset((
    "root", "verbose", "device", "save_folder", "PHYSICS", "TIME", "DT", "FD_ORDER", "FW", "DAMPING", "FPML", "npower",
    "k_max_PML", "FREE_SURF", "QUELLART", "QUELLTYP", "QUELLTYPB", "FC_SPIKE_1", "FC_SPIKE_2", "ORDER_SPIKE", "SEISMO",
    "ITERMAX", "DATA_DIR", "SWS_TAPER_GRAD_HOR", "EXP_TAPER_GRAD_HOR", "GRADT1", "GRADT2", "GRADT3", "GRADT4", "INVMAT1",
    "fwi_stages", "loss", "DT_used"))
# DENISE parameters that cannot change what ONE gradient evaluation (ITERMAX = 1) returns: the MPI decomposition, names of
# files this shim keeps in memory, logging, model bounds and line-search / optimiser settings that only act on model
# updates, grid sizes that come from the Model object.  Accepted; one warning per name says so.
_INERT = frozenset((
    "NPROCX", "NPROCY", "VPUPPERLIM", "VPLOWERLIM", "VSUPPERLIM", "VSLOWERLIM", "RHOUPPERLIM", "RHOLOWERLIM",
    "SEIS_FILE_VX", "SEIS_FILE_VY", "SEIS_FILE_P", "SEIS_FILE_CURL", "SEIS_FILE_DIV", "SEIS_FORMAT", "JACOBIAN", "MFILE",
    "LOG", "LOG_FILE", "MISFIT_LOG_FILE", "INV_MOD_OUT", "INV_MODELFILE", "NX", "NY", "DH", "NT", "SOURCE_FILE", "REC_FILE",
    "SIGNAL_FILE", "GRAD_METHOD", "PCG_BETA", "NLBFGS", "EPS_SCALE", "STEPMAX", "SCALEFAC", "TESTSHOT_START", "TESTSHOT_END",
    "TESTSHOT_INCR", "PRO", "MIN_ITER", "SNAP", "SNAP_FORMAT", "SNAP_FILE", "TSNAP1", "TSNAP2", "TSNAPINC", "IDX", "IDY"))
# parameters that DO change the result and are not built: only their neutral value is accepted
_NEUTRAL = {"TIMEWIN": 0, "TRKILL": 0, "NORMALIZE": 0, "INV_STF": 0, "SPATFILTER": 0, "MODEL_FILTER": 0, "SWS_TAPER_GRAD_VERT": 0,
            "SWS_TAPER_GRAD_SOURCES": 0, "SWS_TAPER_CIRCULAR_PER_SHOT": 0, "SWS_TAPER_FILE": 0, "NDT": 1, "MAXRELERROR": 0,
            "RTMOD": 0, "GRAVITY": 0, "INVMAT": 0, "EPRECOND": 0, "RUN_MULTIPLE_SHOTS": 1, "READREC": 0, "READMOD": 0,
            "TW_IND": 0, "GAMMA": 0, "BOUNDARY": 0}
_warned = set()


class Denise:
    """Stand-in for pyapi_denise.Denise: DENISE's parameters as attributes + forward / grad on the HIP propagator.
    Assigning a name this class does not know raises (a typo must not pass for a parameter); a known parameter that has
    no effect on a single gradient evaluation is accepted with one warning; a parameter that would change the result and
    is not built accepts its neutral value only."""

    def __setattr__(self, name, value):
        if name.startswith("_") or name in _SERVED:
            return object.__setattr__(self, name, value)
        if name in _INERT:
            if name not in _warned:
                _warned.add(name)
                import warnings
                warnings.warn("pyapi_denise shim: %s is accepted and has no effect on forward() / grad() "
                              "(see compat/pyapi_denise.py: _INERT)" % name, stacklevel=2)
            return object.__setattr__(self, name, value)
        if name in _NEUTRAL:
            if value != _NEUTRAL[name]:
                raise MifwiError("%s=%r is not implemented: this parameter changes the gradient and only its neutral "
                                 "value %r is served" % (name, value, _NEUTRAL[name]))
            return object.__setattr__(self, name, value)
        raise AttributeError("pyapi_denise shim: unknown parameter %r (served: %s)" % (name, ", ".join(sorted(
            n for n in _SERVED if n.isupper() or n in ("npower", "k_max_PML", "fwi_stages", "save_folder")))))

    def __init__(self, root=None, verbose=1, device=None):
        self.root = root
        self.verbose = verbose
        self.device = device
        self.save_folder = "./outputs/"
        # the pyapi attributes the reference touches (defaults follow pyapi_denise / DENISE.inp)
        self.PHYSICS = 1
        self.TIME = 6.0
        self.DT = None
        object.__setattr__(self, "NPROCX", 1)          # inert here (no warning for the defaults)
        object.__setattr__(self, "NPROCY", 1)
        # INVMAT1: the parameters the gradients refer to - 1: Vp, Vs, rho; 2: Zp = rho Vp, Zs = rho Vs, rho (networks.py:11025);
        # 3: lambda, mu, rho.  get_fwi_gradients keeps DENISE's file names: "..._vp" holds the first, "..._vs" the second
        self.INVMAT1 = 1
        # the reference never sets FD_ORDER (`#d.FD_ORDER = 4` is commented out, networks.py:10447), so every prop()
        # runs the pyapi_denise default, recorded as 2 in SURVEY.md appendix C: same default here (4 on request)
        self.FD_ORDER = 2
        self.FW = 10
        self.DAMPING = 1500.0
        self.FPML = 10.0
        self.npower = 4.0
        self.k_max_PML = 1.0
        self.FREE_SURF = 1
        self.QUELLART = 1
        self.QUELLTYP = 1
        self.QUELLTYPB = 1
        self.QUELLTYP = 1
        self.FC_SPIKE_1 = -5.0
        self.FC_SPIKE_2 = 15.0
        self.ORDER_SPIKE = 5
        self.SEISMO = 1
        self.ITERMAX = 1
        self.DATA_DIR = None
        for k, v in (("SEIS_FILE_VX", None), ("SEIS_FILE_VY", None), ("VPUPPERLIM", 6000.0), ("VPLOWERLIM", 0.0),
                     ("VSUPPERLIM", 4000.0), ("VSLOWERLIM", 0.0), ("RHOUPPERLIM", 3000.0), ("RHOLOWERLIM", 1000.0)):
            object.__setattr__(self, k, v)
        self.SWS_TAPER_GRAD_HOR = 0
        self.EXP_TAPER_GRAD_HOR = 2.0
        self.GRADT1, self.GRADT2, self.GRADT3, self.GRADT4 = 21, 25, 490, 500
        self.fwi_stages = []
        self._observed = None
        self._gradients = None
        self._gradients_dev = None
        self._shots = None
        self._shots_p = None
        self._observed_p = None
        self.loss = None

    # -- protocol no-ops ------------------------------------------------------------------------
    def set_paths(self, *a, **k):
        return None

    def help(self, *a, **k):
        if self.verbose:
            print("physicsbasedfwi2_amd pyapi_denise shim: attributes are plain Python fields")

    def add_fwi_stage(self, fc_low=0.0, fc_high=0.0, inv_vp_iter=0, inv_vs_iter=0,
                      inv_rho_iter=0, lnorm=2, order=6, **kw):
        stage = dict(fc_low=fc_low, fc_high=fc_high, inv_vp_iter=inv_vp_iter,
                     inv_vs_iter=inv_vs_iter, inv_rho_iter=inv_rho_iter, lnorm=lnorm, order=order)
        stage.update(kw)
        self.fwi_stages.append(stage)

    # -- observed data ----------------------------------------------------------------------------
    def set_observed(self, vx, vy, p=None):
        """Observed particle velocities (and, for QUELLTYPB = 4, pressure), each [nshot, nt, nrec] (array or
        tensor), in shot order of the ``Sources`` passed to :meth:`grad`."""
        self._observed = (torch.as_tensor(np.asarray(vx) if not torch.is_tensor(vx) else vx).float(),
                          torch.as_tensor(np.asarray(vy) if not torch.is_tensor(vy) else vy).float())
        self._observed_p = None if p is None else torch.as_tensor(np.asarray(p) if not torch.is_tensor(p) else p).float()

    def load_observed_su(self, nshots):
        """DENISE layout: DATA_DIR + '_x.su.shot<k>' / '_y.su.shot<k>' (networks.py:7690-7692)."""
        vx, vy = [], []
        for k in range(1, nshots + 1):
            ax, _ = read_su("%s_x.su.shot%d" % (self.DATA_DIR, k))
            ay, _ = read_su("%s_y.su.shot%d" % (self.DATA_DIR, k))
            vx.append(ax.T)
            vy.append(ay.T)
        self.set_observed(np.stack(vx), np.stack(vy))

    # -- numerics -----------------------------------------------------------------------------------
    def _setup(self, model, src, rec):
        if self.PHYSICS not in (1, 2):
            raise MifwiError("PHYSICS=%s not implemented (1 = P-SV elastic, 2 = acoustic)" % self.PHYSICS)
        dev = torch.device(self.device) if self.device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        h = model.dx
        vmax = float(model.vp.max())
        # FD_ORDER (left commented at networks.py:10447): 2 or 4 are built (Taylor weights, MAXRELERROR = 0)
        if int(self.FD_ORDER) not in (2, 4):
            raise MifwiError("FD_ORDER=%s not implemented: the staggered-grid stencils are built for order 2 "
                             "and 4" % self.FD_ORDER)
        limit = profiles.elastic_cfl_limit(h, vmax, int(self.FD_ORDER))
        dt = float(self.DT) if self.DT else _nice_dt(limit)
        if dt > limit:
            raise MifwiError("DT=%g s violates the stability limit %g s (h=%g m, vp_max=%g m/s)"
                             % (dt, limit, h, vmax))
        nt = int(round(self.TIME / dt))
        nz, nx = model.ny, model.nx
        _, _, sc = profiles.cells_round(src.x, src.y, h, nx)
        _, _, rc = profiles.cells_round(rec.x, rec.y, h, nx)
        ns, nr = len(src), len(rec)
        geom = dict(
            sc=torch.tensor(sc).view(ns, 1, 1), sw=torch.ones(ns, 1, 1),
            rc=torch.tensor(rc).view(1, nr, 1).repeat(ns, 1, 1), rw=torch.ones(ns, nr, 1))
        if self.QUELLART == 1:
            wav = np.stack([ricker_denise(src.f[i], nt, dt, src.td[i]) * src.amp[i]
                            for i in range(ns)], axis=1)
        elif self.QUELLART == 3:
            if src.wavelets is None:
                raise MifwiError("QUELLART=3 needs Sources(..., wavelets=[nshot, nt] samples at DT)")
            w = np.broadcast_to(np.atleast_2d(src.wavelets), (ns, np.atleast_2d(src.wavelets).shape[1]))
            wav = np.zeros((nt, ns))
            k = min(nt, w.shape[1])
            wav[:k] = (w[:, :k] * src.amp[:, None]).T
        elif self.QUELLART == 6:
            wav = np.stack([spike_denise(nt, dt, self.FC_SPIKE_1, self.FC_SPIKE_2, self.ORDER_SPIKE,
                                         src.td[i]) * src.amp[i] for i in range(ns)], axis=1)
        else:
            raise MifwiError("QUELLART=%s not implemented (1 Ricker, 3 samples, 6 band-limited spike)"
                             % self.QUELLART)
        # QUELLTYP: the point source (1 explosive, 2 / 3 force along x / y = depth).  QUELLTYPB is DENISE's
        # ADJOINT source type - which seismogram components enter the misfit (1 both, 2 y only, 3 x only,
        # 4 pressure); networks.py:10452 sets 2 for the hydrophone-free real-data case.
        if self.QUELLTYP not in (1, 2, 3):
            raise MifwiError("QUELLTYP=%s not implemented (1 explosive, 2 force x, 3 force y)" % self.QUELLTYP)
        if self.QUELLTYPB not in (1, 2, 3, 4):
            raise MifwiError("QUELLTYPB=%s not implemented (1: x and y components, 2: y only, 3: x only, "
                             "4: pressure)" % self.QUELLTYPB)
        # SEISMO: 1 particle velocities, 2 pressure, 4 both (3 = curl/div is not implemented); the propagator
        # always returns the velocities, pressure on request
        if self.SEISMO not in (1, 2, 4):
            raise MifwiError("SEISMO=%s not implemented (1 velocities, 2 pressure, 4 both)" % self.SEISMO)
        if self.QUELLTYPB == 4 and self.SEISMO == 1:
            raise MifwiError("QUELLTYPB=4 (pressure adjoint sources) needs SEISMO 2 or 4")
        # explosive source: moment-rate density added to sxx and szz; a point force keeps the bare wavelet
        # here and gets dt/(h^2 rho) at the source node from elastic.force_amplitude (differentiable)
        scale = dt / (h * h) if self.QUELLTYP == 1 else 1.0
        f = torch.tensor(wav * scale, dtype=torch.float32).view(nt, ns, 1)
        fw = int(self.FW)
        fsurf = bool(int(self.FREE_SURF))
        pz = torch.tensor(profiles.cpml_tables(nz, fw, h, dt, self.DAMPING, self.FPML, self.npower,
                                               self.k_max_PML, low=not fsurf))
        px = torch.tensor(profiles.cpml_tables(nx, fw, h, dt, self.DAMPING, self.FPML, self.npower,
                                               self.k_max_PML))
        return dev, h, dt, nt, geom, f, pz, px, fw, fsurf

    def _materials(self, model, dev, dt, h, requires_grad, fsurf=False):
        # undo the caller's flipud: internally row 0 is the surface
        # PHYSICS = 2 (acoustic): the same velocity-stress solver in a fluid, Vs = 0 everywhere
        vs = model.vs if self.PHYSICS == 1 else np.zeros_like(model.vs)
        prm = [torch.tensor(np.flipud(a).copy(), device=dev, requires_grad=requires_grad)
               for a in (model.vp, vs, model.rho)]
        return prm, elastic.staggered_materials(prm[0], prm[1], prm[2], dt, h, free_surface=fsurf)

    def _propagate(self, mat, f, pz, px, g, fw, fsurf, h):
        kind = {1: "explosive", 2: "fx", 3: "fz"}[self.QUELLTYP]
        if kind != "explosive":
            f = elastic.force_amplitude(f, mat, g["sc"], g["sw"], h, kind)
        out = elastic.propagate(mat, f, pz, px, g["sc"], g["sw"], g["rc"], g["rw"], fw,
                                free_surface=fsurf, source_type=kind, record_pressure=self.SEISMO in (2, 4),
                                fd_order=int(self.FD_ORDER))
        # DENISE's pressure seismogram: p = -(sxx + syy) at the receiver node
        return (out[0], out[1], -out[2]) if len(out) == 3 else (out[0], out[1], None)

    def forward(self, model, src, rec):
        """Forward modelling; seismograms are kept in memory (``get_shots``)."""
        dev, h, dt, nt, g, f, pz, px, fw, fsurf = self._setup(model, src, rec)
        with torch.no_grad():
            _, mat = self._materials(model, dev, dt, h, False, fsurf)
            vx, vy, p = self._propagate(mat, f.to(dev), pz, px, g, fw, fsurf, h)
        self._shots = (vx.permute(1, 2, 0).cpu().numpy(), vy.permute(1, 2, 0).cpu().numpy())
        self._shots_p = None if p is None else p.permute(1, 2, 0).cpu().numpy()
        self.DT_used = dt
        return self._shots

    def get_shots(self, keys=("_y",), return_filenames=False):
        """List of [nrec, nt] arrays, one per shot, for the component named in ``keys``."""
        if self._shots is None:
            raise MifwiError("run forward() first")
        if any("_p" in k for k in keys):
            if self._shots_p is None:
                raise MifwiError("pressure seismograms need SEISMO = 2 or 4")
            out = [a for a in self._shots_p]
            names = ["su/seis_p.su.shot%d" % (i + 1) for i in range(len(out))]
            return (out, names) if return_filenames else out
        comp = 0 if any("_x" in k for k in keys) else 1
        out = [a for a in self._shots[comp]]
        names = ["su/seis%s.su.shot%d" % ("_x" if comp == 0 else "_y", i + 1) for i in range(len(out))]
        return (out, names) if return_filenames else out

    def grad(self, model, src, rec):
        """One gradient evaluation (ITERMAX=1): forward, stage filter, L2 residual against the
        observed data, exact adjoint, chain rule to (Vp, Vs, rho).  Writes ./loss_curve_grad.out."""
        if self._observed is None:
            if self.DATA_DIR:
                self.load_observed_su(len(src))
            else:
                raise MifwiError("no observed data: call set_observed(vx, vy) or set DATA_DIR")
        if int(self.ITERMAX) != 1:
            raise MifwiError("ITERMAX=%s: grad() is one gradient evaluation (the reference sets ITERMAX = 1 and lets its "
                             "own optimiser update the model)" % self.ITERMAX)
        if int(self.INVMAT1) not in (1, 2, 3):
            raise MifwiError("INVMAT1=%s not implemented (1: Vp, Vs, rho; 2: Zp, Zs, rho; 3: lambda, mu, rho)" % self.INVMAT1)
        dev, h, dt, nt, g, f, pz, px, fw, fsurf = self._setup(model, src, rec)
        prm, mat = self._materials(model, dev, dt, h, True, fsurf)
        vx, vy, p = self._propagate(mat, f.to(dev), pz, px, g, fw, fsurf, h)
        ox, oy = (o.to(dev).permute(1, 0, 2) for o in self._observed)     # -> [nt, ns, nrec]
        if ox.shape != vx.shape:
            raise MifwiError("observed data %s do not match modelled %s (nt, nshot, nrec)"
                             % (tuple(ox.shape), tuple(vx.shape)))
        st = self.fwi_stages[-1] if self.fwi_stages else dict(fc_low=0.0, fc_high=0.0, order=6, lnorm=2)
        fl = lambda a: butterworth(a, dt, st.get("fc_low", 0.0), st.get("fc_high", 0.0),
                                   st.get("order", 6))
        # objective and adjoint sources in one fused pass per component (csrc/mifwi_misfit.hip): lnorm = 2 is the
        # L2 norm every stage of the reference asks for (networks.py:7761, 9863, 10503); 5 is DENISE's
        # global-correlation norm (numbering as in the DENISE manual; DENISE itself is not in the reference tree)
        lnorm = int(st.get("lnorm", 2))
        if lnorm not in (2, 5):
            raise MifwiError("lnorm=%s not implemented (2: L2, 5: global correlation)" % lnorm)
        objective = misfit.l2_half if lnorm == 2 else misfit.global_correlation
        loss = 0.0
        if self.QUELLTYPB in (1, 2):
            loss = loss + objective(fl(vy), fl(oy))
        if self.QUELLTYPB in (1, 3):
            loss = loss + objective(fl(vx), fl(ox))
        if self.QUELLTYPB == 4:
            if self._observed_p is None:
                raise MifwiError("QUELLTYPB=4: pass the observed pressure with set_observed(vx, vy, p=...)")
            loss = loss + objective(fl(p), fl(self._observed_p.to(dev).permute(1, 0, 2)))
        loss.backward()
        self.loss = float(loss.detach())
        with open("loss_curve_grad.out", "w") as fh:
            fh.write("%e\n" % self.loss)
        # re-apply the caller's flipud convention on the way out
        grads = [p.grad.detach() for p in prm]
        if int(self.INVMAT1) != 1:
            # the same objective differentiated with respect to (Zp, Zs, rho) or (lambda, mu, rho): exact chain rule of
            # the change of variables, one launch (csrc/mifwi_materials.hip: mifwi_elastic_gradient_parametrization)
            grads = list(elastic.gradient_parametrization([q.detach() for q in prm], grads, int(self.INVMAT1)))
        if int(self.SWS_TAPER_GRAD_HOR) == 1:
            w = torch.tensor(gradient_taper(model.ny, h, self.GRADT1, self.GRADT2, self.GRADT3, self.GRADT4,
                                            self.EXP_TAPER_GRAD_HOR), device=dev)[:, None]
            grads = [g * w for g in grads]
        self._gradients_dev = torch.stack(grads)           # [vp, vs, rho] (INVMAT1 = 2: Zp, Zs, rho; 3: lambda, mu, rho), row 0 = surface
        gvp, gvs, grho = (np.flipud(g.cpu().numpy()).copy() for g in grads)
        self._gradients = {"rho": grho, "vp": gvp, "vs": gvs}
        self.DT_used = dt
        return self.loss

    def get_fwi_gradients(self, keys=("seis",), return_filenames=False):
        """Arrays in filename order: jacobian/..._rho, ..._vp, ..._vs  =>  [rho, vp, vs]
        (the order models/networks.py:7800-7802 indexes).  With INVMAT1 = 2 / 3 the "vp" and "vs" files hold the
        gradients with respect to Zp, Zs / lambda, mu - DENISE keeps the file names (networks.py:11108-11110)."""
        if self._gradients is None:
            raise MifwiError("run grad() first")
        names = ["jacobian/gradient_seis_rho.bin", "jacobian/gradient_seis_vp.bin",
                 "jacobian/gradient_seis_vs.bin"]
        sel = [n for n in names if all(k in n for k in keys)]
        out = [self._gradients[n.rsplit("_", 1)[1].split(".")[0]] for n in sel]
        return (out, sel) if return_filenames else out


def conditioned_gradients(d, vp, vs, rho, mute_rows=25, rho_factor=0.1, sigma=0.0):
    """What models/networks.py:7799-7862 (9877-9919; 10514-10560 with ``sigma=3, mute_rows=5``) computes from
    ``d.get_fwi_gradients`` on the host - flipud back, zero the top rows, rescale every gradient to
    max(model)/max(gradient), rho x 0.1 - in one device call on the gradients ``d.grad`` left on the GPU
    (csrc/mifwi_gradient.hip).  ``vp, vs, rho``: the models whose maxima set the scale (the reference's
    ``vpst, vsst, rhost``), [nz, nx], any device.  Returns (vp_grad, vs_grad, rho_grad) CUDA float tensors in the
    network's orientation (row 0 = surface), ready for ``fake_Vp.backward(vp_grad)``."""
    from .. import conditioning
    if getattr(d, "_gradients_dev", None) is None:
        raise MifwiError("run grad() first")
    dev = d._gradients_dev.device
    models = torch.stack([torch.as_tensor(np.ascontiguousarray(m) if isinstance(m, np.ndarray) else m)
                          .to(device=dev, dtype=torch.float32) for m in (vp, vs, rho)])
    out = conditioning.condition_gradients(d._gradients_dev, models, None, sigma, False, mute_rows,
                                           (1.0, 1.0, rho_factor))
    return out[0], out[1], out[2]


def gradients_allreduce(d, group=None):
    """Shot-parallel use: every rank calls ``d.grad`` on its block of sources, then this sums the
    three gradients and the loss over ranks with ONE all-reduce (physicsbasedfwi2_amd.dist)."""
    from .. import dist as mdist
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    ts = [torch.tensor(d._gradients[k], device=dev) for k in ("rho", "vp", "vs")]
    ts, loss = mdist.all_reduce_gradient(ts, d.loss, group)
    for k, t in zip(("rho", "vp", "vs"), ts):
        d._gradients[k] = t.cpu().numpy()
    d.loss = float(loss)
    return d
