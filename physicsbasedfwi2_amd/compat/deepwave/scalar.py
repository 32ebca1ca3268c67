"""deepwave.scalar.Propagator call protocol (models/networks.py:5408-5411, 5449, 5464):

    prop = deepwave.scalar.Propagator({'vp': model}, dx)      # model [nz, nx] in m/s
    rec = prop(source_amplitudes, x_s, x_r, dt)               # [nt, nshot, nrec]
    loss(rec).backward()                                      # -> model.grad, src.grad

Conventions kept from deepwave: locations are physical units in the model tensor's dimension
order (z, x); cell = trunc(loc/dx); the source term is scaled by vp^2 dt^2; the internal time
step is dt/ceil(dt/dt_max) with band-limited resampling of the wavelet and decimation of the
traces.  Conventions that are this library's own (deepwave's binaries are not available to
pin against): the absorbing layer is the reference's in-tree sponge
(seisgan/fwi/pde/seismic/model.py:6-29), `pml_width` cells wide (default 20), the model is
edge-replicated into it, and rec[n] samples the field before step n.
"""
import functools
import math

import torch
import torch.nn.functional as F

from ... import acoustic, profiles
from ..._lib import MifwiError

DEFAULT_PML_WIDTH = 20
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


@functools.lru_cache(maxsize=32)
def _sponge(n, width, d, h, dt, device):
    """Damping profile of one axis as a device tensor (a new Propagator is built every iteration,
    networks.py:5449: everything that only depends on the geometry is cached)."""
    return torch.from_numpy(profiles.sponge_q(n, width, d, h, dt)).float().to(device)


class Propagator(torch.nn.Module):
    def __init__(self, model, dx, pml_width=None, survey_pad=None, vpmax=None):
        super().__init__()
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

    def forward(self, source_amplitudes, source_locations, receiver_locations, dt):
        vp = self.vp
        if not vp.is_cuda:
            raise MifwiError("model must live on a HIP device: libmifwi has no CPU fallback")
        dev = vp.device
        P = self.pml_width
        dz, dx = self.spacing
        h = min(dz, dx)
        nt = source_amplitudes.shape[0]
        vmax = float(self.vpmax) if self.vpmax is not None else float(vp.detach().max())
        dt_max = CFL_SAFETY * profiles.scalar_cfl_limit(self.spacing, vmax)
        ratio = max(1, int(math.ceil(abs(dt) / dt_max - 1e-9)))
        dti = dt / ratio

        vp_pad = _EdgePad.apply(vp.float(), P)
        n0, n1 = vp_pad.shape
        r = (vp_pad * (dti / h)) ** 2
        f = source_amplitudes.to(device=dev, dtype=torch.float32) * (h * h)
        f = _upsample(f, ratio)
        q0, q1 = _sponge(n0, P, dz, h, dti, str(dev)), _sponge(n1, P, dx, h, dti, str(dev))
        # coordinates -> cells where the coordinates live (no host round trip when they are on the GPU)
        sc, sw = profiles.cells_truncate(source_locations.detach(), self.spacing, P, n1)
        rc, rw = profiles.cells_truncate(receiver_locations.detach(), self.spacing, P, n1)
        rec = acoustic.propagate(r, f, q0, q1, sc, sw, rc, rw, (h / dz) ** 2, (h / dx) ** 2,
                                 shots_per_group=self.shots_per_group, edge_rows=P)
        return rec[::ratio] if ratio > 1 else rec
