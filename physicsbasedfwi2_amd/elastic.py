"""Host side of the elastic (P-SV) propagator: staggered material preparation (differentiable
torch ops, so the chain rule (vp, vs, rho) -> (lambda, mu, averages) is autograd's), plan and
buffer handling, and the ``torch.autograd.Function`` around the HIP forward / adjoint.

Stands where ``d.forward`` / ``d.grad`` of pyapi_denise stand in the reference
(models/networks.py:7752-7802): models in, vx/vz seismograms and Vp/Vs/rho gradients out,
with no mpirun and no files.  All wave arithmetic is in libmifwi.so; there is no CPU path.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import MifwiError
from .acoustic import _Geometry, _require_cuda, _stream

# bytes of snapshot planes + time checkpoints a call may hold (288 GB of HBM per GPU; the rest is left to the caller's
# network and data); MIFWI_EL_SNAPSHOT_BUDGET_GB overrides it (measurements of the checkpointed path on short runs)
DEFAULT_SNAPSHOT_BUDGET = int(float(os.environ.get("MIFWI_EL_SNAPSHOT_BUDGET_GB", "96")) * (1 << 30))


SNAPSHOT_FORMATS = {"f32": _lib.SNAPSHOT_F32, "bf16": _lib.SNAPSHOT_BF16}


def snapshot_mode():
    """Default storage of the five forward snapshot planes between the forward and the adjoint pass:
    "f32" (exact discrete adjoint) unless MIFWI_EL_SNAP=bf16 asks for the compressed planes (see
    :func:`propagate`, ``snapshot_format``)."""
    import os
    v = os.environ.get("MIFWI_EL_SNAP", "f32").lower()
    if v not in SNAPSHOT_FORMATS:
        raise MifwiError("MIFWI_EL_SNAP must be f32 or bf16 (got %r)" % v)
    return v


def snapshot_bytes_per_cell(fmt=None):
    return 10.0 if (fmt or snapshot_mode()) == "bf16" else 20.0


def kernel_family(flags):
    """(forward, adjoint) description of the formulation a plan picked (``plan.layout.kernel_flags``)."""
    fwd = ("single-launch time loop" if flags & _lib.EL_KERNEL_FWD_SINGLE_LAUNCH else
           "one fused V+S launch per step" if flags & _lib.EL_KERNEL_FWD_FUSED_STEP else "one launch per half step")
    adj = ("single-launch time loop" if flags & _lib.EL_KERNEL_ADJ_SINGLE_LAUNCH else
           "one fused S^T+V^T launch per step" if flags & _lib.EL_KERNEL_ADJ_FUSED_STEP else "one launch per half step")
    return fwd, adj


def other_per_step_env(flags=0):
    """(environment, label) selecting the other formulation of the per-step forward: bench.py's in-run
    cross-check runs one shot through both."""
    if flags & _lib.EL_KERNEL_FWD_FUSED_STEP:
        return {"MIFWI_EL_FUSED": "0"}, "one launch per half step"
    return {"MIFWI_EL_FUSED": "1"}, "fused V+S forward launch"


class _MaterialsFn(torch.autograd.Function):
    """staggered_materials on the device in ONE launch each way (csrc/mifwi_materials.hip): the torch expression below
    is ~60 elementwise launches forward and ~100 backward, 2 ms of a 46 ms gradient pass on the reference's 100x300 grid."""

    @staticmethod
    def forward(ctx, vp, vs, rho, s, free_surface):
        lib = _lib.load()
        dev = vp.device
        nz, nx = vp.shape
        ins = [t.detach().to(dtype=torch.float32).contiguous() for t in (vp, vs, rho)]
        out = torch.empty((5, nz, nx), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.mifwi_elastic_materials(dev.index or 0, *[_lib.ptr(t) for t in ins], _lib.ptr(out), nz, nx,
                                                   float(s), int(bool(free_surface)), _stream()))
        ctx.save_for_backward(*ins)
        ctx.s, ctx.free_surface = float(s), int(bool(free_surface))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        vp, vs, rho = ctx.saved_tensors
        dev = vp.device
        nz, nx = vp.shape
        g = g.to(dtype=torch.float32).contiguous()
        grads = [torch.empty_like(vp) for _ in range(3)]
        with torch.cuda.device(dev):
            _lib.check(lib.mifwi_elastic_materials_vjp(dev.index or 0, _lib.ptr(vp), _lib.ptr(vs), _lib.ptr(rho), _lib.ptr(g),
                                                       *[_lib.ptr(t) for t in grads], nz, nx, ctx.s, ctx.free_surface,
                                                       _stream()))
        return grads[0], grads[1], grads[2], None, None


PARAM_VELOCITY, PARAM_IMPEDANCE, PARAM_LAME = 1, 2, 3


def gradient_parametrization(prm, grads, mode):
    """Gradients with respect to (Vp, Vs, rho) -> the parameter set DENISE calls INVMAT1 (``models/networks.py:11025``):
    1 unchanged, 2 (Zp = rho Vp, Zs = rho Vs, rho), 3 (lambda, mu, rho).  ``prm`` = (vp, vs, rho), ``grads`` = the three
    gradients, float32 tensors of one shape on a HIP device; returns three new tensors.  One launch
    (``mifwi_elastic_gradient_parametrization``): the 3 x 3 Jacobian of the change of variables cell by cell."""
    vp, vs, rho = (t.detach().contiguous().float() for t in prm)
    gv, gs, gr = (t.detach().contiguous().float() for t in grads)
    if not vp.is_cuda:
        raise MifwiError("gradient_parametrization: tensors must live on a HIP device (libmifwi has no CPU fallback)")
    out = [torch.empty_like(vp) for _ in range(3)]
    with torch.cuda.device(vp.device):
        _lib.check(_lib.load().mifwi_elastic_gradient_parametrization(
            vp.device.index or 0, int(mode), _lib.ptr(vp), _lib.ptr(vs), _lib.ptr(rho), _lib.ptr(gv), _lib.ptr(gs),
            _lib.ptr(gr), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), vp.numel(), _stream()))
    return tuple(out)


def staggered_materials(vp, vs, rho, dt, h, free_surface=False):
    """[5, nz, nx] = lambda dt/h, (lambda+2mu) dt/h, mu_xz dt/h, dt/(h rho_x), dt/(h rho_z).
    With ``free_surface`` row 0 of the first two planes is put in the effective form the
    stress-imaging condition needs (szz = 0 there): lambda -> 0, lambda+2mu -> (lambda+2mu) -
    lambda^2/(lambda+2mu).

    Arithmetic averaging of density to the vx / vz nodes, harmonic averaging of the shear
    modulus to the sxz node (0 where any of the four is 0, i.e. in water), edge values
    replicated.  Differentiable.  Float32 tensors on a HIP device take the fused kernels of csrc/mifwi_materials.hip (the
    same operations in the same order: identical planes); anything else (CPU tensors of the tests' oracle compositions,
    float64) the plain torch expression below, which is the definition."""
    if (vp.is_cuda and vs.is_cuda and rho.is_cuda and vp.dim() == 2 and vp.shape == vs.shape == rho.shape and
            vp.dtype == vs.dtype == rho.dtype == torch.float32):
        return _MaterialsFn.apply(vp, vs, rho, dt / h, free_surface)
    return _staggered_materials_torch(vp, vs, rho, dt, h, free_surface)


def _staggered_materials_torch(vp, vs, rho, dt, h, free_surface=False):
    mu = rho * vs * vs
    lam = rho * vp * vp - 2.0 * mu
    s = dt / h

    def sh(a, dz, dx):
        a = torch.cat([a, a[-1:, :]], 0)[dz:dz + a.shape[0]] if dz else a
        a = torch.cat([a, a[:, -1:]], 1)[:, dx:dx + a.shape[1]] if dx else a
        return a
    rx = 0.5 * (rho + sh(rho, 0, 1))
    rz = 0.5 * (rho + sh(rho, 1, 0))
    m4 = [mu, sh(mu, 0, 1), sh(mu, 1, 0), sh(mu, 1, 1)]
    anyzero = (m4[0] == 0) | (m4[1] == 0) | (m4[2] == 0) | (m4[3] == 0)
    inv = sum(1.0 / torch.where(m == 0, torch.ones_like(m), m) for m in m4)
    muxz = torch.where(anyzero, torch.zeros_like(mu), 4.0 / inv)
    Ls, Ms = lam * s, (lam + 2.0 * mu) * s
    if free_surface:
        top = torch.zeros_like(Ls)
        top[0] = 1.0
        Ms = Ms - top * (Ls * Ls / Ms)
        Ls = Ls * (1.0 - top)
    return torch.stack([Ls, Ms, muxz * s, s / rx, s / rz])


class ElasticPlan:
    def __init__(self, nz, nx, nt, nshot, nsrc, nrec, ntap, pml_width, device_index,
                 shots_per_group=0, free_surface=0, source_type=0, record_pressure=0, snapshot_format=None,
                 fd_order=4):
        self._lib = _lib.load()
        fmt = SNAPSHOT_FORMATS[snapshot_format or snapshot_mode()]
        self.desc = _lib.ElasticDesc(nz, nx, nt, nshot, nsrc, nrec, ntap, pml_width,
                                     free_surface, shots_per_group, source_type, record_pressure, fmt, fd_order)
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.mifwi_elastic_plan_create(ctypes.byref(self._h), device_index,
                                                       ctypes.byref(self.desc)))
        self.layout = _lib.ElasticLayout()
        _lib.check(self._lib.mifwi_elastic_plan_layout(self._h, ctypes.byref(self.layout)))

    @property
    def handle(self):
        return self._h

    def pass_sizes(self):
        """(forward, adjoint) units per pass of the per-step kernels over the time range: shots / shot groups
        (elastic), shot groups (acoustic) - what stays inside the Infinity Cache."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(self._lib.mifwi_elastic_plan_pass_sizes(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def cluster_slabs(self, adjoint=False):
        """Row slabs per shot of the single-launch time loop (0: one launch per half step)."""
        return int(self._lib.mifwi_elastic_plan_cluster_slabs(self._h, int(bool(adjoint))))

    def close(self):
        if self._h:
            self._lib.mifwi_elastic_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _ArenaLease:
    """Held by the autograd node whose forward wrote the arena's tensor; dies with the node (a graph dropped without a
    backward frees the arena too)."""

    def __init__(self):
        self.released = False


class _SnapshotArena:
    """One snapshot tensor shared by consecutive propagate() calls (gradient_in_shot_chunks): the first call allocates
    it, later ones whose snapshots are not larger take a prefix - no call pays a malloc of >100 GB again, and chunks need
    not be of equal size to hit torch's cached block.  The tensor has ONE owner at a time: a forward leases it, the
    matching backward (or the death of its graph) gives it back; a second forward in between - two components, a
    forward inside a loss closure, retain_graph - does not get it and allocates snapshots of its own, so its
    predecessor's planes are never overwritten."""
    current = None

    def __init__(self):
        self.buf = None
        self._lease = None          # weakref to the _ArenaLease of the forward whose snapshots live in buf

    def __enter__(self):
        self._prev, _SnapshotArena.current = _SnapshotArena.current, self      # re-entering keeps the tensor
        return self

    def __exit__(self, *exc):
        _SnapshotArena.current = self._prev
        self.buf = None
        self._lease = None
        return False

    def busy(self):
        lease = self._lease() if self._lease is not None else None
        return lease is not None and not lease.released

    def lease(self):
        lease = _ArenaLease()
        self._lease = weakref.ref(lease)
        return lease

    def take(self, nt, elems, dev):
        need = nt * elems
        if self.busy() or self.buf is None or self.buf.device != dev or self.buf.numel() < need:
            return None
        return self.buf[:need].view(nt, elems)


class _ElasticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat, f, pz, px, geom, pml_width, shots_per_group, snapshot_budget, free_surface,
                source_type=0, record_pressure=0, snapshot_format=None, fd_order=4):
        _require_cuda(mat, "mat")
        dev = mat.device
        lib = _lib.load()
        _, nz, nx = mat.shape
        nt, ns, nsrc = f.shape
        if geom.src_cell.shape[:2] != (ns, nsrc):
            raise MifwiError("f is [nt,%d,%d] but src_cell is %s" % (ns, nsrc,
                                                                      tuple(geom.src_cell.shape)))
        nrec, ntap = geom.rec_cell.shape[1], geom.rec_cell.shape[2]
        geom.check_cells(nz * nx, "%dx%d" % (nz, nx))
        with torch.cuda.device(dev):
            plan = ElasticPlan(nz, nx, nt, ns, nsrc, nrec, ntap, pml_width, dev.index,
                               shots_per_group, free_surface, source_type, record_pressure, snapshot_format,
                               fd_order)
            lay = plan.layout
            gp = lay.gp
            mat_p = torch.zeros((5, nz, gp), device=dev, dtype=torch.float32)
            mat_p[:, :, :nx] = mat.detach()
            pz_d = pz.to(device=dev, dtype=torch.float32).contiguous()
            px_p = torch.zeros((6, gp), device=dev, dtype=torch.float32)
            px_p[2] = 1.0
            px_p[5] = 1.0
            px_p[:, :nx] = px.to(device=dev, dtype=torch.float32)
            f_d = f.detach().to(dtype=torch.float32).contiguous()
            rvx = torch.empty((nt, ns, nrec), device=dev, dtype=torch.float32)
            rvz = torch.empty((nt, ns, nrec), device=dev, dtype=torch.float32)
            # pressure receivers: sum w (sxx + szz), bound to the plan (mifwi_elastic_plan_bind_pressure)
            rp = torch.empty((nt, ns, nrec) if record_pressure else (0,), device=dev, dtype=torch.float32)
            if record_pressure:
                _lib.check(lib.mifwi_elastic_plan_bind_pressure(plan.handle, _lib.ptr(rp), None))
            work = torch.empty(lay.work_forward_elems, device=dev, dtype=torch.float32)
            need_grad = mat.requires_grad or f.requires_grad
            step_bytes = 4 * lay.snap_step_elems
            seg, snap, ckpt = nt, None, None
            arena = _SnapshotArena.current
            lease = None
            if need_grad and arena is not None:
                snap = arena.take(nt, lay.snap_step_elems, dev)       # memory already held: no budget question
                if snap is not None:
                    lease = arena.lease()
            if need_grad and snap is None:
                # never plan for more than most of the memory that is free right now (other tensors of
                # the training loop share the device); segmentation does not change the results
                snapshot_budget = min(snapshot_budget, int(0.8 * _lib.free_device_bytes(dev)))
                if nt * step_bytes > snapshot_budget:
                    seg = max(1, int(snapshot_budget // (2 * step_bytes)))
                if seg >= nt:
                    seg = nt
                    snap = torch.empty((nt, lay.snap_step_elems), device=dev, dtype=torch.float32)
                    if arena is not None and arena.buf is None:
                        arena.buf = snap.view(-1)
                        lease = arena.lease()
            args = (plan.handle, _lib.ptr(mat_p), _lib.ptr(pz_d), _lib.ptr(px_p), _lib.ptr(f_d),
                    _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w), _lib.ptr(geom.rec_cell),
                    _lib.ptr(geom.rec_w), _lib.ptr(rvx), _lib.ptr(rvz))
            if not need_grad or seg == nt:
                _lib.check(lib.mifwi_elastic_forward(*args, _lib.ptr(snap), _lib.ptr(work), 0, nt,
                                                     _lib.ZERO_STATE, _stream()))
            else:
                ckpt = []
                for b in range(0, nt, seg):
                    if b > 0:
                        ckpt.append(work[:lay.state_elems].clone())
                    _lib.check(lib.mifwi_elastic_forward(*args, None, _lib.ptr(work), b,
                                                         min(b + seg, nt),
                                                         _lib.ZERO_STATE if b == 0 else 0,
                                                         _stream()))
            if need_grad:
                ctx.plan, ctx.geom, ctx.seg, ctx.ckpt, ctx.snap = plan, geom, seg, ckpt, snap
                ctx.lease = lease
                ctx.dims = (nz, nx, nt, ns, nsrc, nrec)
                ctx.need_f = f.requires_grad
                ctx.record_pressure = record_pressure
                ctx.save_for_backward(mat_p, pz_d, px_p, f_d)
            else:
                plan.close()
        return rvx, rvz, rp

    @staticmethod
    def backward(ctx, g_vx, g_vz, g_p=None):
        lib = _lib.load()
        if ctx.plan is None:
            raise MifwiError("backward through the elastic propagator was called twice: the forward snapshots "
                             "(or checkpoints) are freed by the first call - run the forward again")
        mat_p, pz_d, px_p, f_d = ctx.saved_tensors
        plan, geom = ctx.plan, ctx.geom
        lay = plan.layout
        nz, nx, nt, ns, nsrc, nrec = ctx.dims
        dev = mat_p.device
        with torch.cuda.device(dev):
            gx = (torch.zeros((nt, ns, nrec), device=dev) if g_vx is None
                  else g_vx.to(dtype=torch.float32).contiguous())
            gz = (torch.zeros((nt, ns, nrec), device=dev) if g_vz is None
                  else g_vz.to(dtype=torch.float32).contiguous())
            if ctx.record_pressure:
                gp_ = (None if g_p is None or g_p.numel() == 0 else g_p.to(dtype=torch.float32).contiguous())
                # the forward re-runs of a checkpointed backward must not sample again: rec_p unbound
                _lib.check(lib.mifwi_elastic_plan_bind_pressure(plan.handle, None, _lib.ptr(gp_)))
            grad_mat = torch.empty((5, nz, lay.gp), device=dev, dtype=torch.float32)
            grad_f = (torch.zeros((nt, ns, nsrc), device=dev, dtype=torch.float32)
                      if ctx.need_f else None)
            work = torch.empty(lay.work_backward_elems, device=dev, dtype=torch.float32)
            common = (plan.handle, _lib.ptr(mat_p), _lib.ptr(pz_d), _lib.ptr(px_p),
                      _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w), _lib.ptr(geom.rec_cell),
                      _lib.ptr(geom.rec_w), _lib.ptr(gx), _lib.ptr(gz))
            if ctx.snap is not None:
                _lib.check(lib.mifwi_elastic_backward(
                    *common, _lib.ptr(ctx.snap), 0, _lib.ptr(grad_mat), _lib.ptr(grad_f),
                    _lib.ptr(work), nt - 1, 0, _lib.ZERO_STATE | _lib.FINALIZE, _stream()))
            else:
                seg = ctx.seg
                fwork = torch.empty(lay.work_forward_elems, device=dev, dtype=torch.float32)
                snap = torch.empty((seg, lay.snap_step_elems), device=dev, dtype=torch.float32)
                starts = list(range(0, nt, seg))
                first = True
                for si in reversed(range(len(starts))):
                    b, e = starts[si], min(starts[si] + seg, nt)
                    if b == 0:
                        fflags = _lib.ZERO_STATE
                    else:
                        fwork[:lay.state_elems].copy_(ctx.ckpt[si - 1])
                        fflags = 0
                    _lib.check(lib.mifwi_elastic_forward(
                        plan.handle, _lib.ptr(mat_p), _lib.ptr(pz_d), _lib.ptr(px_p),
                        _lib.ptr(f_d), _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w),
                        _lib.ptr(geom.rec_cell), _lib.ptr(geom.rec_w), None, None,
                        _lib.ptr(snap), _lib.ptr(fwork), b, e, fflags, _stream()))
                    flags = (_lib.ZERO_STATE if first else 0) | (_lib.FINALIZE if b == 0 else 0)
                    first = False
                    _lib.check(lib.mifwi_elastic_backward(
                        *common, _lib.ptr(snap), b, _lib.ptr(grad_mat), _lib.ptr(grad_f),
                        _lib.ptr(work), e - 1, b, flags, _stream()))
            plan.close()
            ctx.plan = None
            ctx.snap = None
            ctx.ckpt = None
            if ctx.lease is not None:               # the arena's tensor may serve the next forward
                ctx.lease.released = True
                ctx.lease = None
        return (grad_mat[:, :, :nx].contiguous(), grad_f) + (None,) * 11


def propagate(mat, f, pz, px, src_cell, src_w, rec_cell, rec_w, pml_width,
              shots_per_group=0, snapshot_budget=DEFAULT_SNAPSHOT_BUDGET, free_surface=False,
              source_type="explosive", record_pressure=False, snapshot_format=None, fd_order=4):
    """Elastic forward modelling, differentiable w.r.t. ``mat`` and ``f``.

    mat [5,nz,nx] from :func:`staggered_materials`;  f [nt,nshot,nsrc] (added to sxx and szz);
    source_type "explosive" (DENISE QUELLTYPB 1), or a point force "fx" / "fz" (QUELLTYPB 2 / 3): f is then
    added to vx / vz between the velocity and the stress update - scale it with
    :func:`force_amplitude` so that the density at the source node enters the gradient;
    pz [6,nz], px [6,nx] from :func:`profiles.cpml_tables`;  cells are iz*nx+ix.
    free_surface: row 0 is a stress-free surface (build ``mat`` with ``free_surface=True`` and
    ``pz`` with ``low=False``).
    fd_order: 4 (default) or 2 - spatial order of the staggered first derivatives (DENISE ``FD_ORDER``); the time
    step must respect the order's own stability limit (:func:`profiles.elastic_cfl_limit`).
    snapshot_format: "f32" (default; the gradient is the exact discrete adjoint) or "bf16": the forward snapshot
    planes are kept as bf16 (half the snapshot stream and memory; material gradients within 4e-3 rel-L2 of the
    f32 form in the worst case, seismograms unchanged) on grids that run the per-step kernels; None = MIFWI_EL_SNAP or "f32".
    Returns (rec_vx, rec_vz), each [nt,nshot,nrec], sampled after the velocity update; with
    ``record_pressure`` also rec_p = sum w (sxx + szz) at the receivers after the stress update (DENISE's
    pressure seismogram is ``-rec_p``; such runs use the one-launch-per-half-step kernels)."""
    _require_cuda(mat, "mat")
    geom = _Geometry.get(src_cell, src_w, rec_cell, rec_w, mat.device)
    f = f.to(device=mat.device)
    try:
        st = SOURCE_TYPES[source_type]
    except KeyError:
        raise MifwiError("source_type must be one of %s" % sorted(k for k in SOURCE_TYPES if isinstance(k, str)))
    if snapshot_format is not None and snapshot_format not in SNAPSHOT_FORMATS:
        raise MifwiError("snapshot_format must be one of %s" % sorted(SNAPSHOT_FORMATS))
    rvx, rvz, rp = _ElasticFn.apply(mat, f, pz, px, geom, int(pml_width), int(shots_per_group),
                                    int(snapshot_budget), 1 if free_surface else 0, st,
                                    1 if record_pressure else 0, snapshot_format, int(fd_order))
    return (rvx, rvz, rp) if record_pressure else (rvx, rvz)


SOURCE_TYPES = {"explosive": 0, "fx": 1, "fz": 2, 0: 0, 1: 1, 2: 2}


def resident_shot_chunk(nshot, nt, nz, nx, snapshot_budget=DEFAULT_SNAPSHOT_BUDGET, snapshot_format=None, device=None):
    """How many shots at a time keep the snapshots of ALL `nt` steps inside the budget (and inside 80 % of the memory
    that is free right now): ``nshot`` when everything fits, 0 when not even one shot does.

    :func:`propagate` cuts the time axis into checkpointed segments when the snapshots of a call do not fit, which
    costs one extra forward sweep.  When the misfit is known before the backward pass starts (observed data in hand,
    as in DENISE's ``grad`` or seisgan's ``FWILoss``) the cheaper cut is across SHOTS: they are independent, so a few at
    a time run forward with resident snapshots and straight into their adjoint, gradients adding up
    (:func:`gradient_in_shot_chunks`) - no recomputation at all."""
    per_shot = nt * nz * (4 * ((nx + 3) // 4)) * snapshot_bytes_per_cell(snapshot_format)
    budget = int(snapshot_budget)
    if device is not None and torch.cuda.is_available():
        budget = min(budget, int(0.8 * _lib.free_device_bytes(device)))
    return int(min(nshot, budget // max(per_shot, 1)))


snapshot_arena = _SnapshotArena          # `with elastic.snapshot_arena(): ...` around hand-written chunk loops


def gradient_in_shot_chunks(mat, f, pz, px, src_cell, src_w, rec_cell, rec_w, pml_width, loss_fn, chunk, **kw):
    """Loss and its gradient with the shots taken ``chunk`` at a time (see :func:`resident_shot_chunk`).

    ``loss_fn(rec_vx, rec_vz, shots)`` -> scalar loss of the shots in the slice ``shots`` (the total loss is their
    sum).  d loss / d mat is accumulated into ``mat.grad`` (``mat`` may be a non-leaf: the chain rule runs once, after
    the last chunk), d loss / d f into ``f.grad`` when ``f`` requires it.  Returns the detached total loss.  Other
    keyword arguments go to :func:`propagate`."""
    ns = f.shape[1]
    chunk = max(1, min(int(chunk), ns))
    leaf = mat.detach().requires_grad_(mat.requires_grad)
    total = None
    fgrad = torch.zeros_like(f) if f.requires_grad else None
    with _SnapshotArena():                   # the chunks share the snapshot tensor of the first (largest) one
        for a in range(0, ns, chunk):
            sl = slice(a, min(a + chunk, ns))
            fc = f[:, sl].detach().requires_grad_(f.requires_grad)
            out = propagate(leaf, fc, pz, px, src_cell[sl], src_w[sl], rec_cell[sl], rec_w[sl], pml_width, **kw)
            loss = loss_fn(out[0], out[1], sl)
            loss.backward()
            total = loss.detach() if total is None else total + loss.detach()
            if fgrad is not None:
                fgrad[:, sl] = fc.grad
    if mat.requires_grad:
        if mat.is_leaf:
            mat.grad = leaf.grad if mat.grad is None else mat.grad + leaf.grad
        else:
            mat.backward(leaf.grad)
    if fgrad is not None:
        if f.is_leaf:
            f.grad = fgrad if f.grad is None else f.grad + fgrad
        else:
            f.backward(fgrad)
    return total


def force_amplitude(wavelet, mat, src_cell, src_w, h, source_type):
    """Point-force amplitudes for :func:`propagate`: wavelet [nt,nshot,nsrc] (force per unit length, N/m)
    times dt/(h^2 rho) at the source node - DENISE's ``vx += DT * amp / (DH^2 rho)`` with
    mat[3] = dt/(h rho_x), mat[4] = dt/(h rho_z).  Differentiable w.r.t. ``mat``: the source term's share of
    the density gradient comes from autograd."""
    plane = mat[3 if SOURCE_TYPES[source_type] == 1 else 4].reshape(-1)
    cell = src_cell.to(device=mat.device, dtype=torch.long).clamp_min(0)
    w = src_w.to(device=mat.device, dtype=mat.dtype) * (src_cell.to(mat.device) >= 0)
    b = (plane[cell] * w).sum(dim=-1)                       # [nshot, nsrc]
    return wavelet.to(device=mat.device, dtype=mat.dtype) * (b / h)[None]
