"""deepwave.scalar.Propagator call protocol (models/networks.py:5408-5411, 5449, 5464):

    prop = deepwave.scalar.Propagator({'vp': model}, dx)      # model [nz, nx] in m/s
    rec = prop(source_amplitudes, x_s, x_r, dt)               # [nt, nshot, nrec]
    loss(rec).backward()                                      # -> model.grad, src.grad

Conventions kept from deepwave: locations are physical units in the model tensor's dimension
order (z, x); cell = trunc(loc/dx); the source term is scaled by vp^2 dt^2; the internal time
step is dt/ceil(dt/dt_max) with band-limited resampling of the wavelet and decimation of the
traces; ``pml_width`` is the width of a PML (default 10 cells per side, deepwave's own default - SURVEY.md
appendix C; the reference never passes one: models/networks.py:5408-5411), the model edge-replicated into it.
deepwave's own PML arithmetic cannot be pinned (its binaries are not available): the layer
is the published second-order convolutional PML inside the same scalar scheme (memory variables psi, zeta on the
layer's cells only; exact transposed adjoint; csrc/mifwi_acoustic_cpml.h, oracle/acoustic_cpml.c), and rec[n]
samples the field before step n.  The layer returns 1.1e-3 (10 cells) / 1.6e-4 (20 cells) of the direct wave
(tests/test_acoustic_cpml_oracle.py); inside the model the scheme is the undamped one bit for bit.  Both kernel
families carry it.  ``pml_freq`` (Hz) sets the frequency shift of the layer (default: a fifth of the source band's
upper end, 0.25 / dt / 5); the damping profile scales with the model's maximum velocity rounded UP to the next
50 m/s (or ``vpmax`` when given), so that the layer does not change from one FWI iteration to the next with the
last digits of the model.

``Propagator(..., absorbing="sponge")`` is the opt-out: the reference's in-tree damping layer
(seisgan/fwi/pde/seismic/model.py:6-29) instead of a PML - 2.6-4.3x faster on Marmousi-sized grids (DESIGN.md
section 3), but it returns 4e-2 of the direct wave at 20 cells.  (Until round 3 it was the default.)

``absorbing="cpml-staggered"`` (round 2's C-PML) advances the scalar equation as the first-order pressure-velocity
system on the staggered grid - the P-SV solver of this library in a fluid (Vs = 0, rho = 1) with its C-PML on every
derivative (csrc/mifwi_elastic.hip; DENISE's PHYSICS = 2 runs the same way): u = -(sxx + szz)/2, source = the
stress increment -vp^2 dt^2 cumsum(f).  Same orders of accuracy, a different discretisation (the two agree to the
discretisation error inside the model); several times slower.

``cfl``: the internal time step is dt / ceil(dt / dt_max).  "stability" (default): dt_max = 0.9 x the stability limit
of the fourth-order stencil; "deepwave": deepwave's own bound dt_max = 0.6 / (vp_max sqrt(sum 1/dx_i^2)) (SURVEY.md
appendix C) - 1.4x tighter, so a caller that wants deepwave's sub-step ratio (2 above 4243 m/s at dx = 10 m,
dt = 1 ms) gets it; a number = that fraction of the stability limit.
"""
import functools
import math
import os

import torch
import torch.nn.functional as F

from ... import acoustic, elastic, profiles
from ..._lib import MifwiError

DEFAULT_PML_WIDTH = 10       # deepwave's default; until round 3: 20
# deepwave's own safety factor (0.6/sqrt(sum 1/dx^2)) is tighter than the stability limit of
# the 4th-order stencil; the true limit with a 10 % margin keeps dt = 1 ms, dx = 10 m stable
# up to 4.9 km/s without sub-stepping.
CFL_SAFETY = 0.9


def _spacing(dx, ndim=2):
    if isinstance(dx, torch.Tensor):
        dx = dx.tolist()
    if isinstance(dx, (int, float)):
        return [float(dx)] * ndim
    dx = [float(v) for v in dx]
    if len(dx) != ndim:
        raise MifwiError("dx must have one entry per model dimension")
    return dx


def _upsample(f, ratio):
    """Band-limited (zero-padded spectrum) resampling of [nt, ...] along time; differentiable."""
    if ratio == 1:
        return f
    nt = f.shape[0]
    spec = torch.fft.rfft(f, dim=0)
    n_up = nt * ratio
    out = torch.zeros((n_up // 2 + 1,) + tuple(f.shape[1:]), dtype=spec.dtype, device=f.device)
    out[:spec.shape[0]] = spec
    if nt % 2 == 0:
        out[spec.shape[0] - 1] = out[spec.shape[0] - 1] * 0.5
    return torch.fft.irfft(out, n=n_up, dim=0) * ratio


class _EdgePad(torch.autograd.Function):
    """Edge-replicating pad of a [nz, nx] model by P cells.  torch's replicate-pad backward sums the
    layer's gradient into the edge cells with atomics (run-to-run different rounding); this one folds
    the layer in with ordered slice sums, so the model gradient is bit-for-bit repeatable."""

    @staticmethod
    def forward(ctx, x, P):
        ctx.P = P
        return F.pad(x[None, None], (P, P, P, P), mode="replicate")[0, 0] if P > 0 else x.clone()

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        if P == 0:
            return g, None
        rows = g[P:-P].clone()
        rows[0] += g[:P].sum(dim=0)
        rows[-1] += g[-P:].sum(dim=0)
        out = rows[:, P:-P].clone()
        out[:, 0] += rows[:, :P].sum(dim=1)
        out[:, -1] += rows[:, -P:].sum(dim=1)
        return out, None


# (weakref to the caller's location tensor, its version, spacing, pad, n1) -> (cells, weights) on the device.  The reference
# builds x_s / x_r once and passes them to a new Propagator every iteration (networks.py:5430-5449): the coordinates ->
# cells conversion (a dozen small launches) and, downstream, the validation of the cells are paid once.  Device tensors
# only: a host tensor may alias a numpy buffer, which has no version counter.
_CELLS = []


def _cells(loc, spacing, pad, n1, dev):
    import weakref
    key = (tuple(spacing), int(pad), int(n1))
    nocache = os.environ.get("MIFWI_NO_GEOM_CACHE", "0") not in ("", "0")     # acoustic._GEOMETRIES: what the key cannot see
    if loc.is_cuda and not nocache:
        for ref, ver, k, out in _CELLS:
            if ref() is loc and ver == (loc._version, loc.data_ptr(), tuple(loc.shape)) and k == key:
                return out
    cells, w = profiles.cells_truncate(loc.detach(), spacing, pad, n1)
    out = (cells.to(dev), w.to(dev))
    if loc.is_cuda and not nocache:
        _CELLS[:] = [e for e in _CELLS if e[0]() is not None][-15:]
        _CELLS.append((weakref.ref(loc), (loc._version, loc.data_ptr(), tuple(loc.shape)), key, out))
    return out


class _Coefficients(torch.autograd.Function):
    """r = (edge-replicated vp * dt / h) ** 2 on the padded grid and its chain rule, one launch each way
    (csrc/mifwi_materials.hip: mifwi_acoustic_coefficients) instead of pad, scale, square and their autograd nodes
    (~20 small launches per gradient pass).  The layer's gradient is folded into the edge cells in a fixed order."""

    @staticmethod
    def forward(ctx, vp, pad, scale):
        from ... import _lib
        from ...acoustic import _stream
        dev = vp.device
        nz, nx = vp.shape
        v = vp.detach().contiguous()
        r = torch.empty((nz + 2 * pad, nx + 2 * pad), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mifwi_acoustic_coefficients(dev.index or 0, _lib.ptr(v), _lib.ptr(r), nz, nx, int(pad),
                                                               float(scale), _stream()))
        ctx.save_for_backward(v)
        ctx.pad, ctx.scale = int(pad), float(scale)
        return r

    @staticmethod
    def backward(ctx, g):
        from ... import _lib
        from ...acoustic import _stream
        (v,) = ctx.saved_tensors
        dev = v.device
        nz, nx = v.shape
        g = g.to(dtype=torch.float32).contiguous()
        gv = torch.empty_like(v)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mifwi_acoustic_coefficients_vjp(dev.index or 0, _lib.ptr(v), _lib.ptr(g), _lib.ptr(gv), nz, nx,
                                                                   ctx.pad, ctx.scale, _stream()))
        return gv, None, None


@functools.lru_cache(maxsize=32)
def _sponge(n, width, d, h, dt, device):
    """Damping profile of one axis as a device tensor (a new Propagator is built every iteration,
    networks.py:5449: everything that only depends on the geometry is cached)."""
    return torch.from_numpy(profiles.sponge_q(n, width, d, h, dt)).float().to(device)


@functools.lru_cache(maxsize=32)
def _cpml_ab(n, width, d, dt, vkey, fpml, device):
    """a and b profiles of one axis of the second-order C-PML as a device tensor: [2, n], zero outside the layer.
    `vkey`: the velocity the damping scales with (_pml_velocity): the cache hits from one iteration to the next."""
    return torch.from_numpy(profiles.cpml_tables(n, width, d, dt, vkey, fpml)[:2].copy()).float().to(device)


def _pml_velocity(vmax):
    """Velocity the C-PML damping profile scales with: the model's maximum rounded up to the next 50 m/s."""
    return 50.0 * math.ceil(vmax / 50.0 - 1e-9)


class Propagator(torch.nn.Module):
    def __init__(self, model, dx, pml_width=None, survey_pad=None, vpmax=None, absorbing="cpml", pml_freq=None,
                 cfl="stability"):
        super().__init__()
        if absorbing not in ("sponge", "cpml", "cpml-staggered"):
            raise MifwiError("absorbing must be 'sponge', 'cpml' or 'cpml-staggered'")
        if not (cfl in ("stability", "deepwave") or (isinstance(cfl, (int, float)) and 0 < cfl <= 1)):
            raise MifwiError("cfl must be 'stability', 'deepwave' or a fraction of the stability limit in (0, 1]")
        self.absorbing = absorbing
        self.cfl = cfl
        self.pml_freq = pml_freq                 # C-PML only: dominant frequency (Hz) of the frequency shift
        if not isinstance(model, dict) or "vp" not in model:
            raise MifwiError("model must be a dict holding 'vp'")
        vp = model["vp"]
        if vp.dim() != 2:
            raise MifwiError("only 2-D models [nz, nx] are supported")
        self.vp = vp
        self.spacing = _spacing(dx)
        self.pml_width = DEFAULT_PML_WIDTH if pml_width is None else int(pml_width)
        self.vpmax = vpmax
        self.shots_per_group = 0

    def _dt_max(self, limit, vmax):
        """Largest internal time step for the given stability limit of the scheme in use."""
        if self.cfl == "deepwave":
            return min(limit, 0.6 / (vmax * math.sqrt(sum(1.0 / (h * h) for h in self.spacing))))
        return (CFL_SAFETY if self.cfl == "stability" else float(self.cfl)) * limit

    def _forward_cpml(self, source_amplitudes, source_locations, receiver_locations, dt):
        vp = self.vp
        dev = vp.device
        P = self.pml_width
        dz, dx = self.spacing
        if abs(dz - dx) > 1e-9 * max(dz, dx):
            raise MifwiError("absorbing='cpml' needs square cells (dz == dx)")
        h = dz
        vmax = float(self.vpmax) if self.vpmax is not None else float(vp.detach().max())
        dt_max = self._dt_max(profiles.elastic_cfl_limit(h, vmax, 4), vmax)
        ratio = max(1, int(math.ceil(abs(dt) / dt_max - 1e-9)))
        dti = dt / ratio
        vp_pad = _EdgePad.apply(vp.float(), P)
        nz, nx = vp_pad.shape
        mat = elastic.staggered_materials(vp_pad, torch.zeros_like(vp_pad), torch.ones_like(vp_pad), dti, h)
        sc, sw = profiles.cells_truncate(source_locations.detach(), self.spacing, P, nx)
        rc, rw = profiles.cells_truncate(receiver_locations.detach(), self.spacing, P, nx)
        f = _upsample(source_amplitudes.to(device=dev, dtype=torch.float32), ratio)
        # stress increment whose second time difference is vp^2 dt^2 f at the source cell (differentiable in vp)
        vp_src = vp_pad.reshape(-1)[sc.to(dev).long().clamp(min=0)].reshape(f.shape[1:])
        a = -(dti * dti) * vp_src * vp_src * torch.cumsum(f, dim=0)
        fpml = float(self.pml_freq) if self.pml_freq is not None else 0.25 / abs(dt) / 5.0
        pz = torch.tensor(profiles.cpml_tables(nz, P, h, dti, vmax, fpml))
        px = torch.tensor(profiles.cpml_tables(nx, P, h, dti, vmax, fpml))
        _, _, rp = elastic.propagate(mat, a, pz, px, sc, sw, rc, rw, P, shots_per_group=self.shots_per_group,
                                     record_pressure=True)
        # u = -(sxx + szz)/2 after the stress update of step n is u^{n+1}; rec[n] samples u^n (sponge convention)
        u = torch.cat([torch.zeros_like(rp[:1]), -0.5 * rp[:-1]], dim=0)
        return u[::ratio] if ratio > 1 else u

    def forward(self, source_amplitudes, source_locations, receiver_locations, dt):
        vp = self.vp
        if not vp.is_cuda:
            raise MifwiError("model must live on a HIP device: libmifwi has no CPU fallback")
        if self.absorbing == "cpml-staggered":
            return self._forward_cpml(source_amplitudes, source_locations, receiver_locations, dt)
        dev = vp.device
        P = self.pml_width
        dz, dx = self.spacing
        h = min(dz, dx)
        nt = source_amplitudes.shape[0]
        vmax = float(self.vpmax) if self.vpmax is not None else float(vp.detach().max())
        dt_max = self._dt_max(profiles.scalar_cfl_limit(self.spacing, vmax), vmax)
        ratio = max(1, int(math.ceil(abs(dt) / dt_max - 1e-9)))
        dti = dt / ratio

        n0, n1 = vp.shape[0] + 2 * P, vp.shape[1] + 2 * P
        if vp.dtype == torch.float32:
            r = _Coefficients.apply(vp, P, dti / h)
        else:
            r = (_EdgePad.apply(vp.float(), P) * (dti / h)) ** 2
        f = source_amplitudes.to(device=dev, dtype=torch.float32) * (h * h)
        f = _upsample(f, ratio)
        # coordinates -> cells where the coordinates live (no host round trip when they are on the GPU)
        sc, sw = _cells(source_locations, self.spacing, P, n1, dev)
        rc, rw = _cells(receiver_locations, self.spacing, P, n1, dev)
        if self.absorbing == "cpml" and P > 0:
            fpml = float(self.pml_freq) if self.pml_freq is not None else 0.25 / abs(dt) / 5.0
            vkey = float(self.vpmax) if self.vpmax is not None else _pml_velocity(vmax)
            ab0, ab1 = _cpml_ab(n0, P, dz, dti, vkey, fpml, str(dev)), _cpml_ab(n1, P, dx, dti, vkey, fpml, str(dev))
            rec = acoustic.propagate(r, f, ab0, ab1, sc, sw, rc, rw, (h / dz) ** 2, (h / dx) ** 2,
                                     shots_per_group=self.shots_per_group, cpml_width=P)
            return rec[::ratio] if ratio > 1 else rec
        q0, q1 = _sponge(n0, P, dz, h, dti, str(dev)), _sponge(n1, P, dx, h, dti, str(dev))
        rec = acoustic.propagate(r, f, q0, q1, sc, sw, rc, rw, (h / dz) ** 2, (h / dx) ** 2,
                                 shots_per_group=self.shots_per_group, edge_rows=P)
        return rec[::ratio] if ratio > 1 else rec
