"""ctypes binding of the C ABI in include/epipolar_amd.h.

The product path has NO CPU fallback: if the HIP library is missing or fails to
load, importing this module raises.  (`oracle/` is test infrastructure and is
never imported from here.)
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# EPIPOLAR_AMD_LIB: a development build of the same ABI (profiling counters, scripts/ws_profile.py); unset in production
LIB_PATH = os.environ.get("EPIPOLAR_AMD_LIB") or os.path.join(_PKG, "lib", "libepipolar_amd.so")

ET_CAM_STRIDE = 27
ET_VARIANT_SAFE_REDUCE = 1
ET_VARIANT_NO_TAP_CACHE = 2
ET_VARIANT_PIXEL_INTERLEAVE = 4
ET_VARIANT_BATCH4 = 8
ET_VARIANT_OCC5 = 16
ET_VARIANT_OCC6 = 32
ET_VARIANT_BASELINE = 256
ET_VARIANT_PIPELINE = 512
ET_VARIANT_MULTI2 = 1024
ET_VARIANT_MULTI4 = 2048
ET_VARIANT_BWD_ATOMIC = 4096
ET_VARIANT_BWD_UNSORTED = 8192
ET_VARIANT_NO_TILE = 16384
ET_VARIANT_TILE_SPLIT = 32768
ET_VARIANT_TILE_CLASSIC = 65536
ET_VARIANT_WS_SETPRIO = 262144
ET_VARIANT_TILE_EXACT = 524288
ET_VARIANT_WS_BAND = 1048576
ET_VARIANT_BWD_SPLIT_IN_PLACE = 2097152
ET_ABI_VERSION = 13
ET_GENERAL_POOLING = 1
ET_GENERAL_PRIOR_MUL = 2
ET_GENERAL_COSINE = 4
ET_GENERAL_ATTENTION_MAX = 8
ET_GENERAL_SIM_PRIOR = 16


class EpipolarAmdError(RuntimeError):
    pass


class EtLayerDesc(ctypes.Structure):
    """Mirror of `struct EtLayerDesc` (include/epipolar_amd.h)."""

    _fields_ = [
        ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("xmin", ctypes.c_float), ("ymin", ctypes.c_float), ("xmax", ctypes.c_float), ("ymax", ctypes.c_float),
        ("eps", ctypes.c_float), ("downsample", ctypes.c_float),
        ("image_resize", ctypes.c_float), ("predict_resize", ctypes.c_float),
        ("correct_normalize", ctypes.c_int32), ("align_corners", ctypes.c_int32),
        ("softmax_scale", ctypes.c_float), ("softmax_enabled", ctypes.c_int32),
        ("src_grad_mask", ctypes.c_int32), ("variant", ctypes.c_int32),
    ]


_P = ctypes.c_void_p
_D = ctypes.POINTER(EtLayerDesc)
_SIGNATURES = {
    "et_abi_version": (ctypes.c_int, []),
    "et_last_error": (ctypes.c_char_p, []),
    "et_sample_locs": (ctypes.c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "et_epipolar_forward": (ctypes.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "et_epipolar_forward_general": (ctypes.c_int, [_D] + [_P] * 8 + [ctypes.c_int32] * 3 + [_P] * 4),
    "et_epipolar_backward_general": (ctypes.c_int, [_D] + [_P] * 9 + [ctypes.c_int32] * 3 + [_P] * 5),
    "et_epipolar_forward_workspace_bytes": (ctypes.c_size_t, [_D]),
    "et_epipolar_forward_workspace_stats_offset": (ctypes.c_size_t, [_D]),
    "et_epipolar_forward_workspace_error_offset": (ctypes.c_size_t, [_D]),
    "et_epipolar_forward_tiled": (ctypes.c_int, [_D] + [_P] * 12 + [ctypes.c_size_t, _P]),
    "et_epipolar_forward_fused": (ctypes.c_int, [_D] + [_P] * 12 + [ctypes.c_int32, _P, ctypes.c_size_t, _P]),
    "et_epipolar_backward_tiled_workspace_bytes": (ctypes.c_size_t, [_D]),
    "et_epipolar_backward_tiled": (ctypes.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "et_epipolar_backward_tiled_attn": (ctypes.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "et_epipolar_backward_workspace_bytes": (ctypes.c_size_t, [_D]),
    "et_epipolar_backward": (ctypes.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "et_residual_epilogue": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "et_residual_gemm_packed_bytes": (ctypes.c_size_t, []),
    "et_residual_gemm_pack": (ctypes.c_int, [_P, _P, _P]),
    "et_residual_gemm": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, _P]),
    "et_z_batch_stats_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "et_z_batch_stats": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "et_z_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "et_z_wgrad": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "et_z_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "et_z_backward": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, _P, ctypes.c_int32, _P, _P, _P, _P, _P,
                                     ctypes.c_size_t, _P]),
    "et_heatmap_peaks": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_int32, _P, _P, _P]),
    "et_nchw_to_nhwc": (ctypes.c_int, [ctypes.c_int32] * 4 + [_P, _P, _P]),
    "et_nhwc_to_nchw": (ctypes.c_int, [ctypes.c_int32] * 4 + [_P, _P, _P]),
    "et_debug_host_sample_setup": (ctypes.c_int, [_D, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P, _P]),
    "et_debug_atomic_probe": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _P]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load the in-tree HIP library (raises if it was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EpipolarAmdError(
            "HIP library %s not found. Build it with `python -m epipolar_transformers_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.et_abi_version()
    if ver != ET_ABI_VERSION:
        raise EpipolarAmdError("ABI version mismatch: library %d, binding %d" % (ver, ET_ABI_VERSION))
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().et_last_error()
        raise EpipolarAmdError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
