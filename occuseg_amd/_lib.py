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
    "scn_grid_chain_temp_bytes": (c_size_t, [c_int64, c_int]),
    "scn_grid_chain_counts": (c_int, [_P, c_int, c_int, c_int64, c_int, _P, c_size_t, ctypes.POINTER(c_int32), _P]),
    "scn_grid_chain_fetch": (c_int, [_P, c_int64, c_int, ctypes.POINTER(c_int32), _P]),
    "scn_grid_build_async": (c_int, [_P, c_int, c_int, c_int, c_int64, _P, c_size_t, _P, _P, _P, _P, _P, _P]),
    "scn_grid_finalize": (c_int, [_P, c_int, c_int, c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P]),
    "scn_hash_capacity": (c_int64, [c_int64]),
    "scn_hash_build": (c_int, [_P, c_int64, _P, _P, c_int64, _P]),
    "scn_input_rules_max_active": (c_int, [_P, c_int64, _P, ctypes.POINTER(c_int32), _P]),
    "scn_input_rules_table": (c_int, [_P, _P, c_int64, c_int32, _P, _P]),
    "scn_rules_temp_bytes": (c_size_t, [c_int64, c_int]),
    "scn_submanifold_rules": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_size_t, _P, c_int, c_int, _P, _P]),
    "scn_strided_rules": (c_int, [_P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "scn_tile_bases": (c_int, [_P, _P, c_int, c_int64, c_int, c_int64, _P, _P]),
    "scn_batch_layout_query": (c_int, [_P, _P]),
    "scn_batch_build": (c_int, [_P, _P, _P, c_size_t, _P]),
    "scn_occ_bounds": (c_int, [_P, c_int, c_int, c_int64, _P, ctypes.POINTER(c_int32), _P]),
    "scn_occ_plan": (c_int, [ctypes.POINTER(c_int32), c_int, _P]),
    "scn_occ_chain": (c_int, [_P, c_int, c_int, c_int64, _P, _P, c_size_t, ctypes.POINTER(c_int32), _P]),
    "scn_occ_fetch": (c_int, [_P, _P, ctypes.POINTER(c_int32), _P]),
    "scn_batch_build_occ": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P, c_size_t, _P]),
    "scn_tile_passes_bytes": (c_size_t, [c_int64]),
    "scn_tile_passes": (c_int, [_P, c_int, c_int64, _P, _P]),
    "scn_conv_wpack_bytes": (c_size_t, [c_int, c_int, c_int]),
    "scn_conv_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "scn_conv_plan": (c_int, [c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "scn_weight_pack": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "scn_weight_pack_blocks": (c_int64, [c_int, c_int, c_int]),
    "scn_weight_pack_flags": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "scn_weight_pack_batch": (c_int, [_P, c_int, c_int64, _P]),
    "scn_conv_stats_blocks": (c_int64, [c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "scn_conv_forward": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int,
                                 c_int64, c_int64, c_int, c_int, c_int64, c_int64, _P, c_int, _P, c_size_t, _P, _P]),
    "scn_conv_dweight_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "scn_conv_dweight": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int64,
                                 c_int64, _P, c_size_t, _P, _P]),
    "scn_column_sum": (c_int, [_P, _P, c_int64, c_int, _P]),
    "scn_bn_temp_bytes": (c_size_t, [c_int]),
    "scn_bn_forward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_double, c_double, c_int, c_double,
                               c_int64, c_int, _P, _P]),
    "scn_bn_forward_ext": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_double, c_double, c_double, c_int64, c_int, _P,
                                   c_int, ctypes.POINTER(c_int32), ctypes.POINTER(_P), ctypes.POINTER(c_int32), _P, _P]),
    "scn_bn_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_double, c_int64, c_int, c_int, _P, _P]),
    "scn_bn_backward_ext": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_double, c_int64, c_int, c_int, _P, _P, c_int, _P,
                                    c_int64, _P, _P]),
    "scn_input_forward": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "scn_input_backward": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "scn_output_forward": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "scn_output_backward": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "scn_site_normals": (c_int, [_P, _P, _P, c_int64, _P, _P, _P]),
    "scn_lookup_sites": (c_int, [_P, c_int64, _P, c_int, c_int64, c_int, c_int, _P, _P]),
    "scn_segment_reduce": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P]),
    "scn_segment_count": (c_int, [_P, c_int64, c_int, _P, _P]),
    "scn_segment_gather": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P]),
    "scn_segment_sqdev": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P]),
}

# only in libscn_hip_tuning.so (-DSCN_TUNING): variant selection, timing-only ablation builds, cycle stamps
TUNING_SIGNATURES = {
    "scn_set_tuning": (c_int, [c_int, c_int]),
    "scn_debug_stamps": (c_int, [_P]),
    "scn_peak_mfma_f32": (c_int, [_P, c_int, ctypes.POINTER(c_double), _P]),
    "scn_mfma_probe": (c_int, [_P, _P, c_int, c_int, ctypes.POINTER(c_double), _P]),
    "scn_peak_copy": (c_int, [_P, _P, c_int64, c_int, ctypes.POINTER(c_double), _P]),
    "scn_peak_read": (c_int, [_P, _P, c_int64, c_int, ctypes.POINTER(c_double), _P]),
}
TUNING_LIB_PATH = os.path.join(_HERE, "libscn_hip_tuning.so")
ABI_VERSION = 7
SK_STATE_BYTES = 65536  # SCN_SK_STATE_BYTES
GRID_CHAIN_MAX = 12     # SCN_GRID_CHAIN_MAX
DW_STATE_BYTES = 1048576  # SCN_DW_STATE_BYTES

_lib = None


BATCH_MAX_LEVELS = 12  # SCN_BATCH_MAX_LEVELS


class BatchPlan(ctypes.Structure):
    """scn_batch_plan of include/scn_hip.h"""
    _fields_ = [("levels", c_int32), ("coords_dtype", c_int32), ("ncols", c_int32), ("flags", c_int32), ("P", c_int64),
                ("N", c_int64 * BATCH_MAX_LEVELS), ("sub", c_int32 * BATCH_MAX_LEVELS), ("strided", c_int32 * BATCH_MAX_LEVELS)]


class BatchLayout(ctypes.Structure):
    """scn_batch_layout of include/scn_hip.h"""
    _fields_ = [("total", c_int64), ("scratch", c_int64), ("small", c_int64), ("order", c_int64), ("site_of_row", c_int64),
                ("site_start", c_int64)] + [(n, c_int64 * BATCH_MAX_LEVELS) for n in (
                    "coords", "keys", "hash_keys", "hash_vals", "hash_cap", "sub_rules", "sub_bases", "str_rules", "str_bases_fine",
                    "str_bases_coarse")]


class OccBox(ctypes.Structure):
    """scn_occ_box of include/scn_hip.h"""
    _fields_ = [("levels", c_int32), ("nb", c_int32), ("origin", c_int32 * 4), ("n", (c_int32 * 3) * BATCH_MAX_LEVELS),
                ("wx", c_int32 * BATCH_MAX_LEVELS), ("woff", c_int64 * (BATCH_MAX_LEVELS + 1)), ("blocks", c_int64),
                ("off_info", c_int64), ("off_blk", c_int64), ("off_words", c_int64), ("bytes", c_int64)]


class BnBwdStats(ctypes.Structure):
    """scn_bn_bwd_stats of include/scn_hip.h"""
    _fields_ = [("x", c_void_p), ("save_mean", c_void_p), ("save_invstd", c_void_p), ("weight", c_void_p),
                ("bias", c_void_p), ("leakiness", c_double)]


def _open(path, signatures):
    if not os.path.exists(path):
        raise RuntimeError(
            "occuseg_amd: %s is missing -- build it with `python -m occuseg_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.scn_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("occuseg_amd: %s has ABI version %d, expected %d" % (os.path.basename(path), got, ABI_VERSION))
    return lib


def load():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def use_tuning_build():
    """tools/ and the kernel-variant test only: route every call of this process through libscn_hip_tuning.so
    (the same sources with -DSCN_TUNING: scn_set_tuning / scn_debug_stamps and the ablation instantiations).
    Returns the handle; `use_product_build()` switches back."""
    global _lib
    sigs = dict(SIGNATURES)
    sigs.update(TUNING_SIGNATURES)
    _lib = _open(TUNING_LIB_PATH, sigs)
    return _lib


def use_product_build():
    global _lib
    _lib = None
    return load()


def check(status):
    if status != 0:
        msg = load().scn_last_error()
        raise RuntimeError("libscn_hip: error %d: %s" % (status, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return t.data_ptr() or None  # (an empty tensor's pointer is 0)


def stream():
    """handle of the current HIP stream of the current device.  torch.cuda.current_stream().cuda_stream builds a Stream
    object through three Python layers (9 us; 120 calls per forward pass of a UNet: a quarter of the host time of the
    forward-only configuration); the raw getter is the same lookup without them."""
    return _raw_stream(_cur_device())


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_device = torch._C._cuda_getDevice
except AttributeError:  # (a torch build without the raw getters)
    def _raw_stream(_idx):
        return torch.cuda.current_stream().cuda_stream

    def _cur_device():
        return 0
