"""ctypes binding of libmifwi.so (the C-ABI declared in include/mifwi.h).

There is no CPU fallback: if the shared object is missing or no HIP device is visible the
compute entry points raise ``MifwiError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIFWI_LIB: load another build of the library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("MIFWI_LIB") or os.path.join(_HERE, "libmifwi.so")

MIFWI_OK = 0
ZERO_STATE = 1
FINALIZE = 2
_ERR_NAMES = {-1: "EINVAL", -2: "ENODEVICE", -3: "EHIP", -4: "ECFL", -5: "ENOMEM"}


class MifwiError(RuntimeError):
    pass


class AcousticDesc(ctypes.Structure):
    _fields_ = [("n0", ctypes.c_int32), ("n1", ctypes.c_int32), ("nt", ctypes.c_int32),
                ("nshot", ctypes.c_int32), ("nsrc", ctypes.c_int32), ("nrec", ctypes.c_int32),
                ("ntap", ctypes.c_int32), ("c0", ctypes.c_float), ("c1", ctypes.c_float),
                ("shots_per_group", ctypes.c_int32), ("edge_rows", ctypes.c_int32),
                ("cpml_width", ctypes.c_int32)]


class AcousticLayout(ctypes.Structure):
    _fields_ = [("gp", ctypes.c_int32), ("pitch", ctypes.c_int32), ("ngroups", ctypes.c_int32),
                ("shots_per_group", ctypes.c_int32), ("field_elems", ctypes.c_int64),
                ("coef_elems", ctypes.c_int64), ("work_forward_elems", ctypes.c_int64),
                ("work_backward_elems", ctypes.c_int64), ("state_elems", ctypes.c_int64)]


class ElasticDesc(ctypes.Structure):
    _fields_ = [("nz", ctypes.c_int32), ("nx", ctypes.c_int32), ("nt", ctypes.c_int32),
                ("nshot", ctypes.c_int32), ("nsrc", ctypes.c_int32), ("nrec", ctypes.c_int32),
                ("ntap", ctypes.c_int32), ("pml_width", ctypes.c_int32),
                ("free_surface", ctypes.c_int32), ("shots_per_group", ctypes.c_int32),
                ("source_type", ctypes.c_int32), ("record_pressure", ctypes.c_int32),
                ("snapshot_format", ctypes.c_int32), ("fd_order", ctypes.c_int32)]


class ElasticLayout(ctypes.Structure):
    _fields_ = [("gp", ctypes.c_int32), ("pitch", ctypes.c_int32), ("ngroups", ctypes.c_int32),
                ("shots_per_group", ctypes.c_int32), ("coef_elems", ctypes.c_int64),
                ("state_elems", ctypes.c_int64), ("work_forward_elems", ctypes.c_int64),
                ("work_backward_elems", ctypes.c_int64), ("snap_step_elems", ctypes.c_int64),
                ("snapshot_format", ctypes.c_int32), ("kernel_flags", ctypes.c_int32)]


EL_KERNEL_FWD_SINGLE_LAUNCH, EL_KERNEL_ADJ_SINGLE_LAUNCH, EL_KERNEL_FWD_FUSED_STEP, EL_KERNEL_ADJ_FUSED_STEP = 1, 2, 4, 8
EL_KERNEL_FWD_LANE_HALO, EL_KERNEL_ADJ_LANE_HALO = 16, 32
SNAPSHOT_F32, SNAPSHOT_BF16 = 0, 1
_P = ctypes.c_void_p

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against
# include/mifwi.h so that header and binding cannot drift apart.
SIGNATURES = {
    "mifwi_last_error": (ctypes.c_char_p, []),
    "mifwi_version": (ctypes.c_int, []),
    "mifwi_fallback_count": (ctypes.c_int64, []),
    "mifwi_agent_handoff_count": (ctypes.c_int64, []),
    "mifwi_slow_handoff_count": (ctypes.c_int64, []),
    "mifwi_device_count": (ctypes.c_int, []),
    "mifwi_device_info": (ctypes.c_int, [ctypes.c_int, _P, _P, _P]),
    "mifwi_acoustic_plan_create": (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int,
                                                  ctypes.POINTER(AcousticDesc)]),
    "mifwi_acoustic_plan_destroy": (ctypes.c_int, [_P]),
    "mifwi_acoustic_plan_layout": (ctypes.c_int, [_P, ctypes.POINTER(AcousticLayout)]),
    "mifwi_acoustic_forward": (ctypes.c_int, [_P] * 12 + [ctypes.c_int32] * 3 + [_P]),
    "mifwi_acoustic_backward": (ctypes.c_int, [_P] * 10 + [ctypes.c_int32] + [_P] * 3 +
                                [ctypes.c_int32] * 3 + [_P]),
    "mifwi_elastic_plan_create": (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int,
                                                 ctypes.POINTER(ElasticDesc)]),
    "mifwi_elastic_plan_destroy": (ctypes.c_int, [_P]),
    "mifwi_elastic_plan_layout": (ctypes.c_int, [_P, ctypes.POINTER(ElasticLayout)]),
    "mifwi_elastic_plan_bind_pressure": (ctypes.c_int, [_P, _P, _P]),
    "mifwi_elastic_plan_pass_sizes": (ctypes.c_int, [_P, _P, _P]),
    "mifwi_acoustic_plan_pass_sizes": (ctypes.c_int, [_P, _P, _P]),
    "mifwi_elastic_forward": (ctypes.c_int, [_P] * 13 + [ctypes.c_int32] * 3 + [_P]),
    "mifwi_elastic_backward": (ctypes.c_int, [_P] * 11 + [ctypes.c_int32] + [_P] * 3 +
                               [ctypes.c_int32] * 3 + [_P]),
    "mifwi_acoustic_born": (ctypes.c_int, [_P] * 8 + [ctypes.c_int32] + [_P] * 2 + [ctypes.c_int32] * 3 + [_P]),
    "mifwi_acoustic_plan_cluster_slabs": (ctypes.c_int, [_P, ctypes.c_int32]),
    "mifwi_elastic_plan_cluster_slabs": (ctypes.c_int, [_P, ctypes.c_int32]),
    "mifwi_misfit_work_elems": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64]),
    "mifwi_misfit": (ctypes.c_int, [ctypes.c_int, ctypes.c_int32] + [_P] * 3 + [ctypes.c_int64] * 2 +
                     [_P] * 4),
    "mifwi_gradient_condition_work_elems": (ctypes.c_int64, [ctypes.c_int32]),
    "mifwi_gradient_condition": (ctypes.c_int, [ctypes.c_int] + [_P] * 3 + [ctypes.c_int32] * 3 + [_P, ctypes.c_float] +
                                 [ctypes.c_int32] * 2 + [_P] * 3),
    "mifwi_elastic_materials": (ctypes.c_int, [ctypes.c_int] + [_P] * 4 + [ctypes.c_int32] * 2 + [ctypes.c_float, ctypes.c_int32, _P]),
    "mifwi_elastic_materials_vjp": (ctypes.c_int, [ctypes.c_int] + [_P] * 7 + [ctypes.c_int32] * 2 +
                                    [ctypes.c_float, ctypes.c_int32, _P]),
    "mifwi_elastic_gradient_parametrization": (ctypes.c_int, [ctypes.c_int, ctypes.c_int32] + [_P] * 9 + [ctypes.c_int64, _P]),
    "mifwi_acoustic_coefficients": (ctypes.c_int, [ctypes.c_int] + [_P] * 2 + [ctypes.c_int32] * 3 + [ctypes.c_float, _P]),
    "mifwi_acoustic_coefficients_vjp": (ctypes.c_int, [ctypes.c_int] + [_P] * 3 + [ctypes.c_int32] * 3 + [ctypes.c_float, _P]),
}
MISFIT_L1_TRACE_NORM = 0
MISFIT_L2 = 1
MISFIT_GLOBAL_CORRELATION = 2

_lib = None


def load():
    """Return the loaded library, building it in-tree with hipcc if it is not there yet."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same soname as /opt/rocm's): it must be the one the
    # process loads first, or torch later finds no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        try:
            _build.build()
        except Exception as exc:  # noqa: BLE001
            raise MifwiError(
                "libmifwi.so is missing and could not be built with hipcc (%s); "
                "run `python -m physicsbasedfwi2_amd.build`. There is no CPU fallback." % exc)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != MIFWI_OK:
        msg = load().mifwi_last_error()
        raise MifwiError("libmifwi %s: %s" % (_ERR_NAMES.get(rc, rc),
                                                msg.decode() if msg else "unknown error"))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def free_device_bytes(dev):
    """Memory a new tensor could get on `dev`: what the driver reports free plus what torch's caching allocator holds
    without using it (the snapshot tensor of the previous call, freed a moment ago, sits there - counting only the
    driver's figure made every second gradient pass plan for a fraction of the device)."""
    import torch
    free = torch.cuda.mem_get_info(dev)[0]
    return int(free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
