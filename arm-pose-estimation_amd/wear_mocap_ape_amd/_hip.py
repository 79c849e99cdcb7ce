"""ctypes binding of libape_hip.so (C ABI: include/ape_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile`` into
``arm-pose-estimation_amd/lib/``.  Loading failures are loud: there is no CPU fallback."""
import ctypes as C
import os
from pathlib import Path

# torch FIRST: it loads the HIP runtime (libamdhip64.so.7) that owns the device pointers and streams
# handed to the kernels; libape_hip.so must bind to that same loaded runtime, not open a second copy.
import torch  # noqa: F401

LIB_PATH = Path(os.environ.get("APE_HIP_LIB", Path(__file__).resolve().parents[1] / "lib" / "libape_hip.so"))

APE_OK = 0
LAYOUT_NONE = -1
LAYOUT_ORI_CAL_LARM_UARM_HIPS = 0
LAYOUT_ORI_CAL_LARM_UARM = 1
LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS = 2
F32, F64 = 0, 1
FLAG_NORMALIZE_INPUT = 0x1
FLAG_ALL_STEPS = 0x2
FLAG_DROPOUT_MASKS = 0x4
FLAG_DROPOUT_PHILOX = 0x8
FLAG_PACKED_MSG = 0x20
FLAG_BROADCAST_X = 0x10
FLAG_ANY_PLACEMENT, FLAG_NO_XCD_CLASSES, FLAG_ALT_FORM = 0x08000000, 0x02000000, 0x01000000    # exchange-form selectors (A/B runs, tests)
FLAG_IN_XCD_PLAIN = 0x00400000      # opt-in: plain hand-over stores inside an XCD-pure cluster (the default is write-through)
KERNEL_AUTO, KERNEL_TILE16, KERNEL_CLUSTER, KERNEL_CLUSTER_GEN1, KERNEL_AUTO_GEN1 = 0, 1, 2, 3, 4
PRECISION_F32, PRECISION_F16, PRECISION_F16_GEN1 = 0, 1, 2
MODEL_LSTM, MODEL_FF, MODEL_IMUPOSE = 0, 1, 2
PARSE_WATCH_PHONE_POCKET, PARSE_WATCH_ONLY, PARSE_WATCH_ONLY_PHONE_MSG, PARSE_WATCH_PHONE_UARM = 0, 1, 2, 3
PARSE_SHAPES = {0: (55, 22), 1: (28, 20), 2: (55, 20), 3: (55, 38)}
PARSE_BIG_ENDIAN = 0x100          # OR-ed into a kind: rows are big-endian float32 (the UDP payload as received)
ABI_VERSION = 7

EST_WIDTH = {LAYOUT_ORI_CAL_LARM_UARM_HIPS: 21, LAYOUT_ORI_CAL_LARM_UARM: 14, LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS: 21}
NUM_TARGETS = {LAYOUT_ORI_CAL_LARM_UARM_HIPS: 14, LAYOUT_ORI_CAL_LARM_UARM: 12, LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS: 20}


class ApeDims(C.Structure):
    _fields_ = [("input_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32),
                ("output_size", C.c_int32), ("target_layout", C.c_int32), ("device", C.c_int32),
                ("model_kind", C.c_int32)]


class ApeModelStats(C.Structure):
    _fields_ = [("aborted_checks", C.c_uint64), ("reissued_calls", C.c_uint64), ("lost_calls", C.c_uint64)]


class ApeKalmanDims(C.Structure):
    _fields_ = [("num_ensemble", C.c_int32), ("win_size", C.c_int32), ("device", C.c_int32)]


# every symbol include/ape_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ape_abi_version": (C.c_int, []),
    "ape_last_error": (C.c_char_p, []),
    "ape_device_count": (C.c_int, []),
    "ape_model_create": (C.c_int, [C.POINTER(ApeDims), C.POINTER(C.c_void_p)]),
    "ape_model_destroy": (C.c_int, [C.c_void_p]),
    "ape_model_reserve": (C.c_int, [C.c_void_p, C.c_int32]),
    "ape_model_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ape_weight_blob_floats": (C.c_size_t, [C.POINTER(ApeDims)]),
    "ape_model_set_norm_stats": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_double)] * 4),
    "ape_model_set_body": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ape_lstm_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                   C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "ape_lstm_forward_hs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                      C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ape_fk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "ape_msg_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ape_parse_rows": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "ape_streams_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ape_streams_destroy": (C.c_int, [C.c_void_p]),
    "ape_streams_reset": (C.c_int, [C.c_void_p]),
    "ape_streams_set_mc": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_uint64]),
    "ape_streams_push_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ape_streams_push_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ape_streams_step": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ape_streams_frame_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]),
    "ape_streams_frame_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "ape_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p,
                            C.c_int32, C.c_void_p]),
    "ape_model_set_kernel": (C.c_int, [C.c_void_p, C.c_int32]),
    "ape_model_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "ape_model_check": (C.c_int, [C.c_void_p]),
    "ape_model_recover": (C.c_int, [C.c_void_p]),
    "ape_model_stats": (C.c_int, [C.c_void_p, C.POINTER(ApeModelStats)]),
    "ape_streams_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "ape_streams_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "ape_lstm_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "ape_model_last_kernel": (C.c_char_p, [C.c_void_p]),
    "ape_flops_per_window": (C.c_double, [C.POINTER(ApeDims), C.c_int32]),
    "ape_kalman_create": (C.c_int, [C.POINTER(ApeKalmanDims), C.POINTER(C.c_void_p)]),
    "ape_kalman_destroy": (C.c_int, [C.c_void_p]),
    "ape_kalman_weight_floats": (C.c_size_t, [C.c_void_p]),
    "ape_kalman_noise_floats": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "ape_kalman_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ape_kalman_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p] + [C.c_void_p] * 6),
    "ape_kalman_format_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ape_kalman_check": (C.c_int, [C.c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises ``UserWarning`` (the reference's error convention) if the HIP
    extension has not been built -- the product path never degrades to a CPU implementation."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise UserWarning(f"HIP extension missing: {LIB_PATH} (run `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` or `make -C arm-pose-estimation_amd/csrc`)")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.ape_abi_version() != ABI_VERSION:
            raise UserWarning(f"{LIB_PATH}: ABI version {handle.ape_abi_version()} != {ABI_VERSION}")
        _lib = handle
    return _lib


def check(status: int, what: str = ""):
    """non-zero status -> ``UserWarning`` raised as an exception (nn_models.py:385-400 convention)."""
    if status != APE_OK:
        msg = lib().ape_last_error().decode("utf-8", "replace")
        raise UserWarning(f"[ape_hip] {what}: {msg} (status {status})")


def dptr(array_like, dtype):
    return array_like.ctypes.data_as(C.POINTER(dtype))
