"""ctypes binding of libscn_hip.so (the C ABI of include/scn_hip.h).

There is no CPU fallback: if the library is missing this module raises, and
every SCN entry point refuses non-GPU tensors.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  -- must be imported first: the library binds to torch's HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscn_hip.so")

_P = c_void_p
# name -> (restype, argtypes); mirrors include/scn_hip.h one to one
SIGNATURES = {
    "scn_abi_version": (c_int, []),
    "scn_last_error": (c_char_p, []),
    "scn_grid_build_temp_bytes": (c_size_t, [c_int64]),
    "scn_grid_build": (c_int, [_P, c_int, c_int, c_int, c_int64, _P, c_size_t, _P, _P, _P, _P,
                               ctypes.POINTER(c_int32), _P]),
    "scn_grid_finalize": (c_int, [_P, c_int, c_int, c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P]),
    "scn_hash_capacity": (c_int64, [c_int64]),
    "scn_hash_build": (c_int, [_P, c_int64, _P, _P, c_int64, _P]),
    "scn_input_rules_max_active": (c_int, [_P, c_int64, _P, ctypes.POINTER(c_int32), _P]),
    "scn_input_rules_table": (c_int, [_P, _P, c_int64, c_int32, _P, _P]),
    "scn_rules_temp_bytes": (c_size_t, [c_int64, c_int]),
    "scn_submanifold_rules": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "scn_strided_rules": (c_int, [_P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "scn_tile_bases": (c_int, [_P, _P, c_int, c_int64, c_int, c_int64, _P, _P]),
    "scn_conv_wpack_bytes": (c_size_t, [c_int, c_int, c_int]),
    "scn_conv_forward": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int64,
                                 c_int64, c_int, c_int, _P, _P]),
    "scn_set_tuning": (c_int, [c_int, c_int]),
    "scn_debug_stamps": (c_int, [_P]),
    "scn_conv_dweight": (c_int, [_P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, _P]),
    "scn_column_sum": (c_int, [_P, _P, c_int64, c_int, _P]),
    "scn_bn_temp_bytes": (c_size_t, [c_int]),
    "scn_bn_forward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_double, c_double, c_int, c_double,
                               c_int64, c_int, _P, _P]),
    "scn_bn_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_double, c_int64, c_int, c_int, _P, _P]),
    "scn_input_forward": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "scn_input_backward": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "scn_output_forward": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "scn_output_backward": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
}

_lib = None


def load():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "occuseg_amd: %s is missing -- build it with `python -m occuseg_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.scn_abi_version()
    if got != 1:
        raise RuntimeError("occuseg_amd: libscn_hip.so ABI version %d, expected 1" % got)
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().scn_last_error()
        raise RuntimeError("libscn_hip: error %d: %s" % (status, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return t.data_ptr() if t.numel() else None


def stream():
    return torch.cuda.current_stream().cuda_stream
