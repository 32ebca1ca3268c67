"""``AcousticWaveSolver``-shaped boundary of the reference's seisgan/fwi package
(seisgan/fwi/pde/seismic/acoustic/wavesolver.py:8-209) with the small objects its callers build around it
(``Model`` model.py:32-200, ``TimeAxis`` / ``PointSource`` / ``Receiver`` / ``RickerSource`` source.py:18-231),
served by the HIP acoustic propagator instead of Devito operators:

    model  = Model(origin, spacing, shape, m, nbpml=20)               # m = square slowness [nx, nz] (model.py:182-192)
    src    = RickerSource(name='src', grid=model.grid, f0=0.01, time=time)      # or time_range=TimeAxis(...)
    rec    = Receiver(name='rec', grid=model.grid, ntime=nt, npoint=nrec)
    solver = AcousticWaveSolver(model, source=src, receiver=rec, kernel='OT2', space_order=4)
    rec, u, _      = solver.forward(save=True, m=m0)                  # acoustic_example.py:66, gradient_example.py:103
    grad, _        = solver.gradient(residual, u, m=m0, grad=grad)    # accumulates into grad.data (layers.py:183)
    srca, v, _     = solver.adjoint(rec)                              # acoustic_example.py:80
    drec, u, U, _  = solver.born(dm)                                  # acoustic_example.py:82

Kept from the reference: units (m, ms, km/s, kHz), axis order (x, z), edge padding by ``nbpml``, the damping
field, ``critical_dt``, bilinear sparse points with offset nbpml, Devito's time loop (time = 1..nt-2:
src[time] -> u[time+1], rec[time] <- u[time]; the adjoint runs time = nt-2..1 and samples srca[time] from
v[time]), gradient and Born perturbation on the PADDED grid with every padded cell an independent variable.
``gradient`` returns dPhi/dm for Phi = 0.5 ||rec||^2-style residuals exactly as Devito's ``grad -= u.dt2 * v``
does (tests/test_reference_pins.py pins that identity in fp64).

Sparse data live in host numpy arrays (``.data``, ``.coordinates.data``) as the callers expect; the propagation
runs on the GPU.  The wavefields ``u`` / ``v`` / ``U`` are opaque handles: ``u`` of a ``save=True`` run holds the
device-resident planes the imaging condition needs, not the nt x nx x nz history (``.data`` raises).
Not served: kernel='OT4' (not on the path layers.py:102 takes), space_order other than 4, 3-D models.
"""
import numpy as np
import torch

from .. import acoustic, profiles
from .._lib import MifwiError


class TimeAxis(object):
    """Uniform time axis given by three of start / step / num / stop (the reference's class of this name,
    source.py:18-69); the arithmetic is :func:`physicsbasedfwi2_amd.profiles.time_axis_complete`."""

    def __init__(self, start=None, step=None, num=None, stop=None):
        self.start, self.step, self.num, self.stop = profiles.time_axis_complete(start, step, num, stop)

    def _rebuild(self):
        return TimeAxis(start=self.start, stop=self.stop, num=self.num)

    @property
    def time_values(self):
        return np.linspace(self.start, self.stop, self.num)


class _Data(object):
    """Stand-in for a Devito Function / coordinate holder: ``.data`` is a numpy array."""

    def __init__(self, data):
        self.data = data


class _Grid(object):
    def __init__(self, shape, spacing, origin):
        self.shape, self.spacing, self.origin = tuple(shape), tuple(spacing), tuple(origin)
        self.dim = len(self.shape)
        self.dtype = np.float32


class Function(_Data):
    """``Function(name=, grid=)`` of the callers (the gradient symbol of layers.py:164, gradient_example.py:53)."""

    def __init__(self, name=None, grid=None, **_):
        self.name, self.grid = name, grid
        _Data.__init__(self, np.zeros(grid.shape, dtype=np.float32))


class Model(object):
    def __init__(self, origin, spacing, shape, m, nbpml=20, dtype=np.float32, **tti):
        if len(shape) != 2:
            raise MifwiError("only 2-D models are served (shape %s)" % (tuple(shape),))
        if any(v is not None for v in tti.values()):
            raise MifwiError("TTI parameters are out of scope (SURVEY.md section 2)")
        self.shape = tuple(int(s) for s in shape)
        self.nbpml = int(nbpml)
        self.origin = tuple(origin)
        self._spacing = tuple(float(s) for s in spacing)
        self.grid = _Grid(self.shape_domain, self._spacing, self.origin)
        self.scale = 1.0
        self.m = _Data(np.zeros(self.shape_domain, dtype=np.float32))
        self.vp = m
        damp = (profiles.sponge_profile(self.shape_domain[0], self.nbpml, self._spacing[0])[:, None]
                + profiles.sponge_profile(self.shape_domain[1], self.nbpml, self._spacing[1])[None, :])
        self.damp = _Data(damp.astype(np.float32))

    @property
    def dim(self):
        return 2

    @property
    def spacing(self):
        return self._spacing

    @property
    def dtype(self):
        return np.float32

    @property
    def shape_domain(self):
        return tuple(d + 2 * self.nbpml for d in self.shape)

    @property
    def domain_size(self):
        return tuple((d - 1) * s for d, s in zip(self.shape, self._spacing))

    @property
    def critical_dt(self):
        return profiles.seisgan_critical_dt(self._spacing, self.scale * float(np.max(self._vp)))

    @property
    def vp(self):
        return self._vp

    @vp.setter
    def vp(self, m):
        """Takes SQUARE SLOWNESS, as the reference's setter does (model.py:182-192)."""
        m = np.asarray(m)
        self._vp = 1.0 / np.sqrt(m)
        if m.ndim:
            self.m.data[:] = self.pad(m)
        else:
            self.m.data[:] = float(m)

    def pad(self, data):
        return np.pad(data, [(self.nbpml, self.nbpml)] * len(self.shape), "edge")


class PointSource(object):
    """Sparse points with a time series each.  Accepts both call styles found in the reference:
    ``time_range=TimeAxis`` (source.py:83, layers.py:131,138) and ``ntime=`` / ``time=`` (the examples)."""

    def __init__(self, name=None, grid=None, time_range=None, npoint=None, data=None, coordinates=None,
                 ntime=None, time=None, **_):
        self.name, self.grid = name, grid
        if time_range is not None:
            self._time_range = time_range._rebuild()
            nt = time_range.num
        elif time is not None:
            time = np.asarray(time, dtype=np.float64)
            nt = time.size
            self._time_range = TimeAxis(start=float(time[0]), stop=float(time[-1]), num=int(nt))
        elif ntime is not None:
            nt = int(ntime)
            self._time_range = None
        else:
            raise MifwiError("PointSource needs time_range=, time= or ntime=")
        if npoint is None:
            if coordinates is None:
                raise MifwiError("PointSource needs npoint= or coordinates=")
            npoint = np.asarray(coordinates).shape[0]
        self.npoint = int(npoint)
        self.nt = int(nt)
        self.data = np.zeros((self.nt, self.npoint), dtype=np.float32)
        self.coordinates = _Data(np.zeros((self.npoint, 2), dtype=np.float32))
        if coordinates is not None:
            self.coordinates.data[:] = coordinates
        if data is not None:
            self.data[:] = data

    @property
    def time_range(self):
        return self._time_range

    @property
    def time_values(self):
        return self._time_range.time_values


Receiver = PointSource
Shot = PointSource


class RickerSource(PointSource):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("npoint", 1)
        self.f0 = kwargs.pop("f0")
        PointSource.__init__(self, *args, **kwargs)
        for p in range(self.npoint):
            self.data[:, p] = self.wavelet(self.f0, self.time_values)

    def wavelet(self, f0, t):
        return profiles.ricker_seisgan(f0, t)


class _Wavefield(object):
    """What ``forward`` / ``adjoint`` / ``born`` hand back in the place of a Devito TimeFunction."""

    def __init__(self, name, tape=None):
        self.name = name
        self._tape = tape           # (m_pad leaf, syn tensor) of a save=True forward run

    @property
    def data(self):
        raise MifwiError("the %r wavefield stays on the device as the planes the imaging condition reads; its "
                         "time history is not exported" % self.name)


class AcousticWaveSolver(object):
    def __init__(self, model, source, receiver, kernel="OT2", space_order=2, device=None, **kwargs):
        if kernel != "OT2":
            raise MifwiError("kernel %r: only the second-order time stepping of layers.py:102 is served" % (kernel,))
        if int(space_order) != 4:
            raise MifwiError("space_order %r: the HIP stencil is the fourth-order one the reference's FWI layer "
                             "uses (layers.py:102)" % (space_order,))
        self.model, self.source, self.receiver = model, source, receiver
        self.space_order, self.kernel = int(space_order), kernel
        self.dt = model.critical_dt
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        self._kwargs = kwargs

    # -- shared set-up --------------------------------------------------------------------------------------------
    def _static(self, dt, src_coords, rec_coords):
        mdl = self.model
        p, h = mdl.nbpml, mdl.spacing
        n0, n1 = mdl.shape_domain
        href = min(h)
        c0, c1 = (href / h[0]) ** 2, (href / h[1]) ** 2
        q0 = torch.from_numpy(profiles.sponge_q(n0, p, h[0], href, dt)).float()
        q1 = torch.from_numpy(profiles.sponge_q(n1, p, h[1], href, dt)).float()
        sc, sw = profiles.cells_bilinear(np.asarray(src_coords, dtype=np.float64)[None], h, p, (n0, n1))
        rc, rw = profiles.cells_bilinear(np.asarray(rec_coords, dtype=np.float64)[None], h, p, (n0, n1))
        return href, c0, c1, q0, q1, sc, sw, rc, rw

    def _m_pad(self, m):
        """Square slowness on the padded grid as a float32 device tensor: a Function-like (.data), an array of the
        padded or of the physical shape (edge-padded then), or a scalar."""
        mdl = self.model
        if m is None:
            m = mdl.m
        a = np.asarray(getattr(m, "data", getattr(m, "value", m)), dtype=np.float32)
        if a.ndim == 0:
            a = np.full(mdl.shape_domain, float(a), dtype=np.float32)
        elif a.shape == tuple(mdl.shape):
            a = mdl.pad(a)
        elif a.shape != tuple(mdl.shape_domain):
            raise MifwiError("m has shape %s, expected %s or %s" % (a.shape, mdl.shape, mdl.shape_domain))
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _source_term(self, src, nt, href):
        """Devito's loop injects src[time] into u[time+1] for time = 1..nt-2: f[n] = src[n+1] h^2, n = 0..nt-3."""
        f = np.zeros((nt, 1, src.npoint), dtype=np.float32)
        f[:nt - 2, 0, :] = np.asarray(src.data, dtype=np.float32)[1:nt - 1] * np.float32(href * href)
        return torch.from_numpy(f).to(self.device)

    def _run(self, m_pad, f, st, dt):
        href, c0, c1, q0, q1, sc, sw, rc, rw = st
        r = (dt * dt / (href * href)) / m_pad
        rec = acoustic.propagate(r, f, q0, q1, sc, sw, rc, rw, c0, c1, edge_rows=self.model.nbpml)
        syn = torch.zeros_like(rec)
        nt = rec.shape[0]
        syn[1:nt - 1] = rec[0:nt - 2]                    # rec_devito[time] = u[time]
        return syn

    # -- the four operations --------------------------------------------------------------------------------------
    def forward(self, src=None, rec=None, u=None, m=None, save=None, **kwargs):
        src = src or self.source
        dt = kwargs.pop("dt", self.dt)
        if rec is None:
            rec = Receiver(name="rec", grid=self.model.grid, ntime=self.receiver.nt,
                           coordinates=self.receiver.coordinates.data)
        nt = src.nt
        st = self._static(dt, src.coordinates.data, rec.coordinates.data)
        f = self._source_term(src, nt, st[0])
        m_pad = self._m_pad(m)
        if save:
            m_pad.requires_grad_(True)
            syn = self._run(m_pad, f, st, dt)
            u = _Wavefield("u", tape=(m_pad, syn))
        else:
            with torch.no_grad():
                syn = self._run(m_pad, f, st, dt)
            u = _Wavefield("u")
        rec.data[:] = syn.detach()[:, 0, :].cpu().numpy()
        return rec, u, None

    def gradient(self, rec, u, v=None, grad=None, m=None, **kwargs):
        """rec.data = the data residual; u from ``forward(save=True)`` with the same m.  Adds dPhi/dm (padded grid)
        to ``grad.data`` as Devito's operator accumulates over shots (layers.py:169-183)."""
        if getattr(u, "_tape", None) is None:
            raise MifwiError("gradient() needs the wavefield of a forward(save=True) run")
        m_pad, syn = u._tape
        u._tape = None                                   # the planes are consumed (one gradient per forward run)
        if grad is None:
            grad = Function(name="grad", grid=self.model.grid)
        res = torch.from_numpy(np.ascontiguousarray(rec.data, dtype=np.float32)).to(self.device)[:, None, :]
        (g,) = torch.autograd.grad(syn, m_pad, res)
        grad.data[:] += g.cpu().numpy()
        return grad, None

    def adjoint(self, rec, srca=None, v=None, m=None, **kwargs):
        """rec.data acts as the adjoint source; srca.data[time] samples the adjoint field at the source points."""
        dt = kwargs.pop("dt", self.dt)
        src = self.source
        if srca is None:
            srca = PointSource(name="srca", grid=self.model.grid, ntime=src.nt, coordinates=src.coordinates.data)
        nt = src.nt
        st = self._static(dt, srca.coordinates.data, rec.coordinates.data)
        href = st[0]
        f = torch.zeros((nt, 1, srca.npoint), dtype=torch.float32, device=self.device, requires_grad=True)
        syn = self._run(self._m_pad(m), f, st, dt)
        g = torch.from_numpy(np.ascontiguousarray(rec.data, dtype=np.float32)).to(self.device)[:, None, :]
        (gf,) = torch.autograd.grad(syn, f, g)
        # <rec, J src> = <srca, src> with src[time] = f[time-1] / h^2  =>  srca[time] = gf[time-1] h^2
        out = np.zeros((nt, srca.npoint), dtype=np.float32)
        out[1:nt - 1] = (gf[0:nt - 2, 0, :] * (href * href)).cpu().numpy()
        srca.data[:] = out
        return srca, _Wavefield("v"), None

    def born(self, dmin, src=None, rec=None, u=None, U=None, m=None, **kwargs):
        """Linearised data for the square-slowness perturbation ``dmin`` (padded or physical shape)."""
        src = src or self.source
        dt = kwargs.pop("dt", self.dt)
        if rec is None:
            rec = Receiver(name="rec", grid=self.model.grid, ntime=self.receiver.nt,
                           coordinates=self.receiver.coordinates.data)
        nt = src.nt
        st = self._static(dt, src.coordinates.data, rec.coordinates.data)
        href, c0, c1, q0, q1, sc, sw, rc, rw = st
        f = self._source_term(src, nt, href)
        m_pad = self._m_pad(m)
        dm = np.asarray(getattr(dmin, "data", dmin), dtype=np.float32)
        if dm.shape == tuple(self.model.shape):
            dm = np.pad(dm, [(self.model.nbpml, self.model.nbpml)] * 2, "constant")
        dm = torch.from_numpy(np.ascontiguousarray(dm)).to(self.device)
        k = dt * dt / (href * href)
        r = k / m_pad
        dr = -k / (m_pad * m_pad) * dm                   # r = k / m
        _, drec = acoustic.born(r, f, dr, q0, q1, sc, sw, rc, rw, c0, c1)
        out = np.zeros((nt, rec.npoint), dtype=np.float32)
        out[1:nt - 1] = drec[0:nt - 2, 0, :].cpu().numpy()
        rec.data[:] = out
        return rec, _Wavefield("u"), _Wavefield("U"), None
