"""Fused data misfit + adjoint source on the GPU (csrc/mifwi_misfit.hip through the C-ABI).

Drop-in for the torch expressions the reference evaluates between the propagator call and
``.backward()``:

* ``l1_trace_normalized(pred, obs_norm, direct)``  -- models/networks.py:5467-5476, 5491
  (``obs_norm`` is the observed data already normalised per trace, networks.py:5418-5419)
* ``l2_half(pred, obs)`` -- ``0.5 * sum((pred - obs)**2)``, seisgan/fwi/layers.py:176-178 and
  DENISE's lnorm = 2 objective (networks.py:7758)

* ``global_correlation(pred, obs)`` -- ``-sum_traces <pred, obs> / (|pred| |obs|)``, DENISE's global-correlation
  norm (the ``lnorm`` argument of ``add_fwi_stage``, models/networks.py:9863, 10503)

All return a 0-d tensor that back-propagates into ``pred`` (one kernel pass produces the loss and
dloss/dpred).  No CPU fallback: CPU tensors raise.
"""
import torch

from . import _lib


class _MisfitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, obs, direct, kind):
        if not pred.is_cuda:
            raise _lib.MifwiError("misfit needs CUDA/HIP tensors (libmifwi has no CPU fallback)")
        if pred.dim() < 2 or obs.shape != pred.shape or (direct is not None and direct.shape != pred.shape):
            raise ValueError("pred, obs (and direct) must share one [nt, ...] shape")
        lib = _lib.load()
        p = pred.detach().contiguous().float()
        o = obs.detach().contiguous().float()
        d = None if direct is None else direct.detach().contiguous().float()
        nt = p.shape[0]
        ntrace = p.numel() // nt
        need_adj = pred.requires_grad
        adj = torch.empty_like(p) if need_adj else None
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        work = torch.empty(lib.mifwi_misfit_work_elems(kind, nt, ntrace), device=p.device, dtype=torch.float32)
        _lib.check(lib.mifwi_misfit(p.device.index or 0, kind, _lib.ptr(p), _lib.ptr(o), _lib.ptr(d), nt,
                                    ntrace, _lib.ptr(loss), _lib.ptr(adj), _lib.ptr(work),
                                    torch.cuda.current_stream(p.device).cuda_stream))
        ctx.adj = adj
        return loss

    @staticmethod
    def backward(ctx, g):
        # the adjoint source stays with the graph (a second backward through a retained graph gets it again)
        return (None if ctx.adj is None else ctx.adj * g), None, None, None


def l1_trace_normalized(pred, obs_norm, direct=None):
    """mean |(pred - direct) / (max_t |pred - direct| + 1e-10) - obs_norm|, time on axis 0."""
    return _MisfitFn.apply(pred, obs_norm, direct, _lib.MISFIT_L1_TRACE_NORM)


def l2_half(pred, obs):
    """0.5 * sum((pred - obs)**2)."""
    return _MisfitFn.apply(pred, obs, None, _lib.MISFIT_L2)


def global_correlation(pred, obs):
    """-sum over traces of <pred, obs> / (|pred| |obs|), time on axis 0 (insensitive to trace amplitudes)."""
    return _MisfitFn.apply(pred, obs, None, _lib.MISFIT_GLOBAL_CORRELATION)
