"""ctypes loader for libaai_hip.so (the C ABI declared in include/aai.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C area_average_interpolation_amd/csrc``.
There is no Python or CPU fallback: if the shared object is missing, importing the compute API raises.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# (tools/ may point AAI_LIB at the experiments build, `make -C csrc exp` -> libaai_hip_exp.so, whose launch heuristics read the
# AAI_* environment switches; the product library ignores them)
LIB_PATH = os.environ.get("AAI_LIB") or os.path.join(_HERE, "libaai_hip.so")

# status codes (include/aai.h)
OK = 0
ERR_RESOLUTION_MISMATCH = 1
ERR_RESOLUTION_NONPOSITIVE = 2
ERR_NO_ROWS = 3
ERR_NO_COLUMNS = 4
ERR_NONFINITE = 5
ERR_BAD_ARGUMENT = 6
ERR_TOO_LARGE = 7
ERR_NO_DEVICE = 8
ERR_HIP = 9
ERR_EMPTY_OUTPUT = 10

MODE_AREA, MODE_FAST, MODE_BILINEAR, MODE_BICUBIC = 1, 2, 3, 4
POLICY_REFERENCE, POLICY_EXACT = 0, 1
POLICY_DOUBLE_PRECISION = 0x100      # OR into a policy: general rotations in double precision throughout (include/aai.h)
POLICY_PREFER_CELL = 0x200           # OR into a policy: the cell formulation also for small outputs (a hint)
POLICY_DIAG_NO_FIXUP = 0x400         # OR into a policy: DIAGNOSTIC, the fix-up pass over the plan's listed pixels is not launched
DTYPE_F32, DTYPE_U8, DTYPE_U16 = 0, 1, 2
KERNEL_AXIS, KERNEL_ROTATED, KERNEL_FAST, KERNEL_SAMPLE, KERNEL_AXIS_WIDE = 1, 2, 3, 4, 5


class Request(ctypes.Structure):
    """struct aai_request"""
    _fields_ = [("mode", ctypes.c_int32), ("policy", ctypes.c_int32),
                ("src_width", ctypes.c_int32), ("src_height", ctypes.c_int32),
                ("src_res_x", ctypes.c_double), ("src_res_y", ctypes.c_double),
                ("dst_res_x", ctypes.c_double), ("dst_res_y", ctypes.c_double),
                ("src_iso_x", ctypes.c_double), ("src_iso_y", ctypes.c_double),
                ("rotation_deg", ctypes.c_double)]


class Layout(ctypes.Structure):
    """struct aai_layout"""
    _fields_ = [("dst_width", ctypes.c_int32), ("dst_height", ctypes.c_int32),
                ("dst_iso_x", ctypes.c_double), ("dst_iso_y", ctypes.c_double),
                ("scale", ctypes.c_int32), ("quadrant", ctypes.c_int32),
                ("reduced_angle_deg", ctypes.c_double), ("side", ctypes.c_double),
                ("kernel", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# every symbol include/aai.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
_RQ = ctypes.POINTER(Request)
_LY = ctypes.POINTER(Layout)
_I64 = ctypes.c_int64
SYMBOLS = {
    "aai_query": (ctypes.c_int, [_RQ, _LY]),
    "aai_last_error": (ctypes.c_char_p, []),
    "aai_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "aai_version": (ctypes.c_int, []),
    "aai_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "aai_set_device": (ctypes.c_int, [ctypes.c_int]),
    "aai_device_synchronize": (ctypes.c_int, []),
    "aai_resample_f32": (ctypes.c_int, [_RQ, _P, _I64, _P, _I64, _LY]),
    "aai_resample_f64": (ctypes.c_int, [_RQ, _P, _I64, _P, _I64, _LY]),
    "aai_resample_device_f32": (ctypes.c_int, [_RQ, _P, _I64, _P, _I64, _P]),
    "aai_resample_batch_device_f32": (ctypes.c_int, [_RQ, ctypes.c_int32, _P, _I64, _I64, _P, _I64, _I64, _P]),
    "aai_resample_batch_multi_device_f32": (ctypes.c_int, [_RQ, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                                          ctypes.POINTER(_P), _I64, _I64, ctypes.POINTER(_P), _I64, _I64, ctypes.POINTER(_P)]),
    "aai_resample_batch_device": (ctypes.c_int, [_RQ, ctypes.c_int32, _P, ctypes.c_int32, _I64, _I64, _P, _I64, _I64, _P]),
    "aai_resample_host": (ctypes.c_int, [_RQ, _P, ctypes.c_int32, _I64, _P, _I64, _LY]),
    "aai_resample_interleaved_device": (ctypes.c_int, [_RQ, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, _I64, _I64, _P, _I64, _I64, _P]),
    "aai_resample_interleaved_host": (ctypes.c_int, [_RQ, ctypes.c_int32, _P, ctypes.c_int32, _I64, _P, _I64, _LY]),
    "aai_host_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64]),
    "aai_host_free": (ctypes.c_int, [_P]),
    "aai_resample_batch_host": (ctypes.c_int, [_RQ, ctypes.c_int32, _P, ctypes.c_int32, _I64, _I64, _P, _I64, _I64, _LY]),
    "aai_band_source_rows": (ctypes.c_int, [_RQ, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "aai_resample_band_device_f32": (ctypes.c_int, [_RQ, ctypes.c_int32, ctypes.c_int32, _P, _I64, _P, _I64, _P]),
    "aai_synth_device_f32": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _I64, ctypes.c_uint64, _P]),
    "aai_synth_rows_device_f32": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _I64, ctypes.c_uint64, _P]),
    "aai_prepare": (ctypes.c_int, [_RQ, ctypes.c_int32]),
    "aai_plan_info": (ctypes.c_int, [_RQ, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]),
    "aai_shutdown": (ctypes.c_int, []),
    "aai_last_kernel": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load libaai_hip.so, failing loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libaai_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the resampling path)" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, and a process that loads the system's copy
        # first (through this library) and torch's afterwards finds "No HIP GPUs" in torch.  Loaded after torch, this
        # library binds to the copy already in the process (same SONAME) -- so bring torch in first when it is installed.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the ABI and the header drifted apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
