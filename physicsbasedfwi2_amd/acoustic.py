"""Host side of the acoustic propagator: plan handling, buffer allocation (torch = device
memory + stream plumbing only) and the ``torch.autograd.Function`` through which the model
gradient reaches the caller, as deepwave's autograd backward does at
models/networks.py:5464/5491 and ``FWILoss`` does at seisgan/fwi/layers.py:158-197.

All arithmetic happens in libmifwi.so (HIP); there is no CPU path here.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import MifwiError

# snapshots kept resident between forward and backward; above this the time axis is cut
# into checkpointed segments that are re-propagated during the adjoint (exact, ~1 extra forward)
DEFAULT_SNAPSHOT_BUDGET = 96 << 30


class AcousticPlan:
    """RAII wrapper of ``mifwi_acoustic_plan`` (include/mifwi.h)."""

    def __init__(self, n0, n1, nt, nshot, nsrc, nrec, ntap, c0, c1, device_index,
                 shots_per_group=0, edge_rows=0, cpml_width=0):
        self._lib = _lib.load()
        self.desc = _lib.AcousticDesc(n0, n1, nt, nshot, nsrc, nrec, ntap, c0, c1,
                                      shots_per_group, int(edge_rows), int(cpml_width))
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.mifwi_acoustic_plan_create(ctypes.byref(self._h), device_index,
                                                        ctypes.byref(self.desc)))
        self.layout = _lib.AcousticLayout()
        _lib.check(self._lib.mifwi_acoustic_plan_layout(self._h, ctypes.byref(self.layout)))

    @property
    def handle(self):
        return self._h

    def pass_sizes(self):
        """(forward, adjoint) units per pass of the per-step kernels over the time range: shots / shot groups
        (elastic), shot groups (acoustic) - what stays inside the Infinity Cache."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(self._lib.mifwi_acoustic_plan_pass_sizes(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def cluster_slabs(self, adjoint=False):
        """Row slabs per shot of the single-launch time loop (0: one launch per step)."""
        return int(self._lib.mifwi_acoustic_plan_cluster_slabs(self._h, int(bool(adjoint))))

    def close(self):
        if self._h:
            self._lib.mifwi_acoustic_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _require_cuda(t, name):
    if not t.is_cuda:
        raise MifwiError("%s must live on a HIP device (got %s): libmifwi has no CPU fallback"
                         % (name, t.device))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Geometries already built from the caller's four tap tensors, when those live on the device (weak references + versions:
# an entry is used only while the very same, unmodified tensor objects are passed again).  A training loop passes the same
# acquisition every iteration: the validation of the cells (a host round trip that stalls the launch queue) is paid once.
# What the key can see: the tensor OBJECT, its version counter, its storage address and shape.  What it cannot see: a
# write that bypasses autograd's version counter (`t.data[...] = `, a kernel of another library writing through the raw
# pointer) - a caller that edits an acquisition in place that way must pass a new tensor, or set MIFWI_NO_GEOM_CACHE=1
# (every call then rebuilds and re-validates its geometry).  Negative cells are inactive taps by convention (the
# kernels skip them), so the validation bounds the cells from above only.
_GEOMETRIES = []


def _geom_key(t):
    return (t._version, t.data_ptr(), tuple(t.shape))


class _Geometry:
    """Device-resident sparse-point description shared by forward and backward."""

    @classmethod
    def get(cls, src_cell, src_w, rec_cell, rec_w, device):
        import weakref
        given = (src_cell, src_w, rec_cell, rec_w)
        # host tensors may alias numpy buffers (no version counter there): rebuilt every call
        if not all(t.is_cuda for t in given) or os.environ.get("MIFWI_NO_GEOM_CACHE", "0") not in ("", "0"):
            return cls(src_cell, src_w, rec_cell, rec_w, device)
        for refs, versions, dev, geom in _GEOMETRIES:
            if dev == device and all(r() is t for r, t in zip(refs, given)) and versions == tuple(_geom_key(t) for t in given):
                return geom
        geom = cls(src_cell, src_w, rec_cell, rec_w, device)
        _GEOMETRIES[:] = [e for e in _GEOMETRIES if all(r() is not None for r in e[0])][-15:]
        _GEOMETRIES.append((tuple(weakref.ref(t) for t in given), tuple(_geom_key(t) for t in given), device, geom))
        return geom

    def check_cells(self, ncell, what):
        """Every tap inside the grid (an out-of-grid cell would fault the kernels).  One host round trip, once per
        geometry."""
        if self._top is None:
            tops = [c.max() for c in (self.src_cell, self.rec_cell) if c.numel()]
            self._top = int(torch.stack(tops).max()) if tops else -1
        if self._top >= ncell:
            raise MifwiError("src_cell/rec_cell hold a cell outside the %s grid" % what)

    def __init__(self, src_cell, src_w, rec_cell, rec_w, device):
        self._top = None
        self.src_cell = src_cell.to(device=device, dtype=torch.int32).contiguous()
        self.src_w = src_w.to(device=device, dtype=torch.float32).contiguous()
        self.rec_cell = rec_cell.to(device=device, dtype=torch.int32).contiguous()
        self.rec_w = rec_w.to(device=device, dtype=torch.float32).contiguous()
        if self.src_cell.dim() != 3 or self.rec_cell.dim() != 3:
            raise MifwiError("src_cell/rec_cell must be [nshot, npoint, ntap]")
        if self.src_cell.shape != self.src_w.shape or self.rec_cell.shape != self.rec_w.shape:
            raise MifwiError("cell/weight shape mismatch")
        if self.src_cell.shape[0] != self.rec_cell.shape[0]:
            raise MifwiError("source and receiver shot counts differ")
        if self.src_cell.shape[2] != self.rec_cell.shape[2]:
            raise MifwiError("sources and receivers must use the same number of taps")


class _AcousticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, f, q0, q1, geom, c0, c1, shots_per_group, snapshot_budget, edge_rows, cpml_width=0):
        _require_cuda(r, "r")
        dev = r.device
        lib = _lib.load()
        n0, n1 = r.shape
        nt, ns, nsrc = f.shape
        if geom.src_cell.shape[:2] != (ns, nsrc):
            raise MifwiError("f is [nt,%d,%d] but src_cell is %s" % (ns, nsrc,
                                                                      tuple(geom.src_cell.shape)))
        nrec, ntap = geom.rec_cell.shape[1], geom.rec_cell.shape[2]
        ncell = n0 * n1
        geom.check_cells(ncell, "%dx%d" % (n0, n1))
        with torch.cuda.device(dev):
            plan = AcousticPlan(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1, dev.index,
                                shots_per_group, edge_rows, cpml_width)
            lay = plan.layout
            gp = lay.gp
            r_p = torch.zeros((n0, gp), device=dev, dtype=torch.float32)
            r_p[:, :n1] = r.detach()
            q0_d = q0.to(device=dev, dtype=torch.float32).contiguous()
            if cpml_width > 0:                   # q0 / q1 carry the layer's a, b profiles: [2, n0], [2, n1] -> [2, gp]
                if tuple(q0.shape) != (2, n0) or tuple(q1.shape) != (2, n1):
                    raise MifwiError("cpml_width > 0: q0 / q1 must be the [2, n0] / [2, n1] C-PML profiles (a, b)")
                q1_p = torch.zeros((2, gp), device=dev, dtype=torch.float32)
                q1_p[:, :n1] = q1.to(device=dev, dtype=torch.float32)
            else:
                q1_p = torch.zeros(gp, device=dev, dtype=torch.float32)
                q1_p[:n1] = q1.to(device=dev, dtype=torch.float32)
            f_d = f.detach().to(dtype=torch.float32).contiguous()
            rec = torch.empty((nt, ns, nrec), device=dev, dtype=torch.float32)
            work = torch.empty(lay.work_forward_elems, device=dev, dtype=torch.float32)
            need_grad = r.requires_grad or f.requires_grad
            step_bytes = 4 * ns * lay.coef_elems
            seg = nt
            snap = None
            ckpt = None
            if need_grad:
                # never plan for more than most of the memory that is free right now (other tensors of
                # the training loop share the device); segmentation does not change the results
                snapshot_budget = min(snapshot_budget, int(0.8 * _lib.free_device_bytes(dev)))
                if nt * step_bytes > snapshot_budget:
                    seg = max(1, int(snapshot_budget // (2 * step_bytes)))
                if seg >= nt:
                    seg = nt
                    snap = torch.empty((nt, ns, n0, gp), device=dev, dtype=torch.float32)
            args = (plan.handle, _lib.ptr(r_p), _lib.ptr(q0_d), _lib.ptr(q1_p), _lib.ptr(f_d),
                    _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w), _lib.ptr(geom.rec_cell),
                    _lib.ptr(geom.rec_w), _lib.ptr(rec))
            if not need_grad or seg == nt:
                _lib.check(lib.mifwi_acoustic_forward(*args, _lib.ptr(snap), _lib.ptr(work), 0, nt,
                                                      _lib.ZERO_STATE, _stream()))
            else:
                # checkpoint the state (two time levels + C-PML memory variables) at every segment start, no snapshots yet
                ckpt = []
                state_elems = lay.state_elems
                for b in range(0, nt, seg):
                    flags = _lib.ZERO_STATE if b == 0 else 0
                    if b > 0:
                        ckpt.append(work[:state_elems].clone())
                    _lib.check(lib.mifwi_acoustic_forward(*args, None, _lib.ptr(work), b,
                                                          min(b + seg, nt), flags, _stream()))
            if need_grad:
                ctx.plan = plan
                ctx.geom = geom
                ctx.seg = seg
                ctx.ckpt = ckpt
                ctx.snap = snap
                ctx.dims = (n0, n1, nt, ns, nsrc, nrec)
                ctx.need_f = f.requires_grad
                ctx.save_for_backward(r_p, q0_d, q1_p, f_d)
            else:
                plan.close()
        return rec

    @staticmethod
    def backward(ctx, grad_rec):
        lib = _lib.load()
        if ctx.plan is None:
            raise MifwiError("backward through the acoustic propagator was called twice: the forward snapshots "
                             "(or checkpoints) are freed by the first call - run the forward again")
        r_p, q0_d, q1_p, f_d = ctx.saved_tensors
        plan, geom = ctx.plan, ctx.geom
        lay = plan.layout
        n0, n1, nt, ns, nsrc, nrec = ctx.dims
        dev = r_p.device
        with torch.cuda.device(dev):
            g = grad_rec.to(dtype=torch.float32).contiguous()
            grad_r = torch.empty((n0, lay.gp), device=dev, dtype=torch.float32)
            grad_f = (torch.zeros((nt, ns, nsrc), device=dev, dtype=torch.float32)
                      if ctx.need_f else None)
            work = torch.empty(lay.work_backward_elems, device=dev, dtype=torch.float32)
            common = (plan.handle, _lib.ptr(r_p), _lib.ptr(q0_d), _lib.ptr(q1_p),
                      _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w), _lib.ptr(geom.rec_cell),
                      _lib.ptr(geom.rec_w), _lib.ptr(g))
            if nt < 2:
                grad_r.zero_()
            elif ctx.snap is not None:
                _lib.check(lib.mifwi_acoustic_backward(
                    *common, _lib.ptr(ctx.snap), 0, _lib.ptr(grad_r), _lib.ptr(grad_f),
                    _lib.ptr(work), nt - 1, 1, _lib.ZERO_STATE | _lib.FINALIZE, _stream()))
            else:
                seg = ctx.seg
                fwork = torch.empty(lay.work_forward_elems, device=dev, dtype=torch.float32)
                snap = torch.empty((seg, ns, n0, lay.gp), device=dev, dtype=torch.float32)
                state_elems = lay.state_elems
                starts = list(range(0, nt, seg))
                first = True
                for si in reversed(range(len(starts))):
                    b, e = starts[si], min(starts[si] + seg, nt)
                    # snapshots G^b..G^{e-1} serve adjoint steps k = e .. b+1
                    k_hi, k_lo = min(e, nt - 1), b + 1
                    if k_hi < k_lo:
                        continue
                    if b == 0:
                        fflags = _lib.ZERO_STATE
                    else:
                        fwork[:state_elems].copy_(ctx.ckpt[si - 1])
                        fflags = 0
                    _lib.check(lib.mifwi_acoustic_forward(
                        plan.handle, _lib.ptr(r_p), _lib.ptr(q0_d), _lib.ptr(q1_p), _lib.ptr(f_d),
                        _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w), _lib.ptr(geom.rec_cell),
                        _lib.ptr(geom.rec_w), None, _lib.ptr(snap), _lib.ptr(fwork), b, e, fflags,
                        _stream()))
                    flags = (_lib.ZERO_STATE if first else 0) | (_lib.FINALIZE if b == 0 else 0)
                    first = False
                    _lib.check(lib.mifwi_acoustic_backward(
                        *common, _lib.ptr(snap), b, _lib.ptr(grad_r), _lib.ptr(grad_f),
                        _lib.ptr(work), k_hi, k_lo, flags, _stream()))
            plan.close()
            ctx.plan = None
            ctx.snap = None
            ctx.ckpt = None
        return (grad_r[:, :n1].contiguous(), grad_f, None, None, None, None, None, None, None, None, None)


def propagate(r, f, q0, q1, src_cell, src_w, rec_cell, rec_w, c0=1.0, c1=1.0,
              shots_per_group=0, snapshot_budget=DEFAULT_SNAPSHOT_BUDGET, edge_rows=0, cpml_width=0):
    """Run the acoustic propagator (differentiable w.r.t. ``r`` and ``f``).

    cpml_width = W > 0: the absorbing layer is a second-order convolutional PML of W cells on every side instead of
    the sponge; q0 [2, n0] and q1 [2, n1] then hold its a and b profiles (``profiles.cpml_tables(...)[:2]``), zero
    outside the layer.  Same call otherwise; runs on the one-launch-per-step kernels.

    r   [n0,n1]  = vp^2 dt^2 / h^2 on the computational (already padded) grid
    f   [nt,nshot,nsrc] source amplitudes (the injected term is  w * f[n] * r[cell])
    q0  [n0], q1 [n1]  separable damping,  q = damp h^2 / (2 dt)
    src_cell/src_w [nshot,nsrc,ntap], rec_cell/rec_w [nshot,nrec,ntap]  (cell = i0*n1+i1)
    edge_rows: optional hint, rows of absorbing layer at the top/bottom of the grid (performance only)
    returns rec [nt,nshot,nrec] with rec[n] sampled from u^n.
    """
    _require_cuda(r, "r")
    geom = _Geometry.get(src_cell, src_w, rec_cell, rec_w, r.device)
    f = f.to(device=r.device)
    ntap = geom.src_cell.shape[2]
    if ntap > 1 and not cpml_width and _flatten_taps_pays(r, f, geom, c0, c1, edge_rows):
        # Bilinear taps (the Devito-shaped protocol) as independent single-cell points: every tap becomes
        # a source / receiver of its own, which makes the single-launch time loop eligible; the taps of a
        # point are recombined by differentiable torch ops (sum over the tap axis, repeat of f).
        ns, nsrc = geom.src_cell.shape[:2]
        nrec = geom.rec_cell.shape[1]
        flat = _Geometry(geom.src_cell.reshape(ns, nsrc * ntap, 1), geom.src_w.reshape(ns, nsrc * ntap, 1),
                         geom.rec_cell.reshape(ns, nrec * ntap, 1), geom.rec_w.reshape(ns, nrec * ntap, 1),
                         r.device)
        rec = _AcousticFn.apply(r, f.repeat_interleave(ntap, dim=2), q0, q1, flat, float(c0), float(c1),
                                int(shots_per_group), int(snapshot_budget), int(edge_rows))
        return rec.reshape(rec.shape[0], ns, nrec, ntap).sum(dim=3)
    return _AcousticFn.apply(r, f, q0, q1, geom, float(c0), float(c1), int(shots_per_group),
                             int(snapshot_budget), int(edge_rows), int(cpml_width))


def _flatten_taps_pays(r, f, geom, c0, c1, edge_rows):
    """True when the flattened (single-tap) problem runs on the single-launch kernels' fast paths."""
    import os
    if os.environ.get("MIFWI_AC_FLATTEN_TAPS", "1") == "0":
        return False
    ns, nsrc, ntap = geom.src_cell.shape
    nrec = geom.rec_cell.shape[1]
    if nrec * ntap > 1024 or nsrc * ntap > 64:
        return False
    n0, n1 = r.shape
    with torch.cuda.device(r.device):
        plan = AcousticPlan(n0, n1, f.shape[0], ns, nsrc * ntap, nrec * ntap, 1, float(c0), float(c1),
                            r.device.index, 0, int(edge_rows))
        ok = plan.cluster_slabs() > 0
        plan.close()
    return ok


def born(r, f, dr, q0, q1, src_cell, src_w, rec_cell, rec_w, c0=1.0, c1=1.0,
         snapshot_budget=DEFAULT_SNAPSHOT_BUDGET, cpml_width=0):
    """Born / linearised modelling (``AcousticWaveSolver.born``, wavesolver.py:174-209;
    ``BornOperator``, operators.py:168-207): returns ``(rec, drec)`` where ``rec`` are the seismograms
    of the background model ``r`` and ``drec = J dr`` their first-order change for the perturbation
    ``dr`` (same parametrisation and shape as ``r``).  ``J`` is the exact transpose partner of the
    gradient autograd returns for :func:`propagate`.  No autograd through this call."""
    _require_cuda(r, "r")
    dev = r.device
    lib = _lib.load()
    geom = _Geometry.get(src_cell, src_w, rec_cell, rec_w, dev)
    n0, n1 = r.shape
    nt, ns, nsrc = f.shape
    nrec, ntap = geom.rec_cell.shape[1], geom.rec_cell.shape[2]
    if tuple(dr.shape) != (n0, n1):
        raise MifwiError("dr must have the shape of r")
    geom.check_cells(n0 * n1, "%dx%d" % (n0, n1))
    with torch.cuda.device(dev), torch.no_grad():
        plan = AcousticPlan(n0, n1, nt, ns, nsrc, nrec, ntap, float(c0), float(c1), dev.index, 0, 0, int(cpml_width))
        lay = plan.layout
        gp = lay.gp
        if 4 * nt * ns * lay.coef_elems > snapshot_budget:
            plan.close()
            raise MifwiError("born() keeps the forward snapshots resident: %d steps do not fit the budget" % nt)
        r_p = torch.zeros((n0, gp), device=dev, dtype=torch.float32)
        r_p[:, :n1] = r.detach()
        dr_p = torch.zeros((n0, gp), device=dev, dtype=torch.float32)
        dr_p[:, :n1] = dr.detach().to(dev)
        q0_d = q0.to(device=dev, dtype=torch.float32).contiguous()
        if cpml_width:
            q1_p = torch.zeros((2, gp), device=dev, dtype=torch.float32)
            q1_p[:, :n1] = q1.to(device=dev, dtype=torch.float32)
        else:
            q1_p = torch.zeros(gp, device=dev, dtype=torch.float32)
            q1_p[:n1] = q1.to(device=dev, dtype=torch.float32)
        f_d = f.detach().to(device=dev, dtype=torch.float32).contiguous()
        rec = torch.empty((nt, ns, nrec), device=dev, dtype=torch.float32)
        drec = torch.empty((nt, ns, nrec), device=dev, dtype=torch.float32)
        work = torch.empty(lay.work_forward_elems, device=dev, dtype=torch.float32)
        snap = torch.empty((nt, ns, n0, gp), device=dev, dtype=torch.float32)
        _lib.check(lib.mifwi_acoustic_forward(plan.handle, _lib.ptr(r_p), _lib.ptr(q0_d), _lib.ptr(q1_p),
                                              _lib.ptr(f_d), _lib.ptr(geom.src_cell), _lib.ptr(geom.src_w),
                                              _lib.ptr(geom.rec_cell), _lib.ptr(geom.rec_w), _lib.ptr(rec),
                                              _lib.ptr(snap), _lib.ptr(work), 0, nt, _lib.ZERO_STATE, _stream()))
        _lib.check(lib.mifwi_acoustic_born(plan.handle, _lib.ptr(r_p), _lib.ptr(q0_d), _lib.ptr(q1_p),
                                           _lib.ptr(dr_p), _lib.ptr(geom.rec_cell), _lib.ptr(geom.rec_w),
                                           _lib.ptr(snap), 0, _lib.ptr(drec), _lib.ptr(work), 0, nt,
                                           _lib.ZERO_STATE, _stream()))
        plan.close()
    return rec, drec
