"""ctypes binding of libreid_hip.so (C ABI: include/reid_hip.h).

``cffi`` is not installed in the target image; ctypes speaks the same C ABI.
There is no CPU fallback: if the shared library is missing the import of this
module raises, and so does every product entry point built on it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreid_hip.so")
LIB_PATH = os.environ.get("REID_HIP_LIB", LIB_PATH)   # A/B experiments: an alternative build of the same library

METRIC_L2, METRIC_L2SQR, METRIC_COS_HALF, METRIC_COS, METRIC_DOT = range(5)
COMM_ID_BYTES = 128
K_CONV_GEMM, K_DIST_GEMM, K_ELEMENTWISE, K_SELECT = range(4)

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_SIGS = {
    "reid_last_error": (C.c_char_p, []),
    "reid_device_count": (_i, [C.POINTER(_i)]),
    "reid_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "reid_ctx_destroy": (_i, [_vp]),
    "reid_ctx_set_stream": (_i, [_vp, _vp]),
    "reid_ctx_set_null_stream": (_i, [_vp]),
    "reid_ctx_sync": (_i, [_vp]),
    "reid_device_sync": (_i, [_vp]),
    "reid_ctx_clear_fault": (_i, [_vp]),
    "reid_ctx_set_chunk": (_i, [_vp, _i]),
    "reid_ctx_set_precision": (_i, [_vp, _i]),
    "reid_ctx_precision_ok": (_i, [_vp, _i, _i]),
    "reid_ctx_fault_peek": (_i, [_vp, C.POINTER(_i)]),
    "reid_ctx_set_side_index": (_i, [_vp, _vp, _i]),
    "reid_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "reid_free": (_i, [_vp, _vp]),
    "reid_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "reid_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "reid_timer_start": (_i, [_vp]),
    "reid_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "reid_profile_enable": (_i, [_vp, _i]),
    "reid_profile_reset": (_i, [_vp]),
    "reid_profile_get": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "reid_seres18_load": (_i, [_vp, _vp, _sz, C.c_char_p]),
    "reid_seres18_dims": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "reid_embed_u8": (_i, [_vp, _vp, _i, _vp, _vp]),
    "reid_embed_u8_dev": (_i, [_vp, _vp, _i, _vp, _vp]),
    "reid_embed_ragged_u8": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "reid_embed_frame_u8": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "reid_embed_f32_nchw": (_i, [_vp, _vp, _i, _vp, _vp]),
    "reid_embed_f32_nchw_dev": (_i, [_vp, _vp, _i, _vp, _vp]),
    "reid_swin_load": (_i, [_vp, _vp, _sz, C.c_char_p]),
    "reid_swin_dims": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "reid_swin_embed_f32_nchw": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "reid_swin_embed_f32_nchw_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "reid_ctx_set_debug_keep": (_i, [_vp, _i]),
    "reid_debug_stage": (_i, [_vp, _i, _vp, _sz, C.POINTER(_sz)]),
    "reid_debug_swin_stage": (_i, [_vp, _i, _vp, _sz, C.POINTER(_sz)]),
    "reid_distmat": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "reid_distmat_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "reid_argmin_rows": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "reid_argmin_rows_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "reid_knn": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "reid_knn_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "reid_descriptor_f32_nchw": (_i, [_vp, _vp, _i, _i, _vp]),
    "reid_descriptor_f32_nchw_dev": (_i, [_vp, _vp, _i, _i, _vp]),
    "reid_cam_debias": (_i, [_vp, _vp, _vp, _i, _i, C.c_float, _i]),
    "reid_cam_debias_dev": (_i, [_vp, _vp, _vp, _i, _i, C.c_float, _i]),
    "reid_smooth_tracklets": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_float]),
    "reid_smooth_tracklets_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_float]),
    "reid_bank_create": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "reid_bank_destroy": (_i, [_vp]),
    "reid_bank_update": (_i, [_vp, _vp, _vp, _vp, _i]),
    "reid_bank_update_dev": (_i, [_vp, _vp, _vp, _vp, _i]),
    "reid_bank_clear": (_i, [_vp, _vp, _vp, _i]),
    "reid_bank_count": (_i, [_vp, _i, C.POINTER(_i)]),
    "reid_bank_cost": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, C.c_float, _vp]),
    "reid_bank_cost_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, C.c_float, _vp]),
    "reid_host_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "reid_host_free": (_i, [_vp, _vp]),
    "reid_frame_submit": (_i, [_vp, _i, _vp, _vp, _vp, _i]),
    "reid_frame_cost": (_i, [_vp, _i, _vp, _vp, _i, _i, C.c_float, _vp, _vp, _i]),
    "reid_frame_cost_groups": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, C.c_float, _vp, _vp, _i]),
    "reid_frame_match_stream": (_i, [_vp, _i]),
    "reid_frame_fetch": (_i, [_vp, _i, _vp, _vp, _vp]),
    "reid_frame_gather": (_i, [_vp, _i, _i]),
    "reid_frame_update": (_i, [_vp, _i, _vp, _vp, _vp, _i]),
    "reid_rerank_jaccard": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "reid_rerank_jaccard_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "reid_diou": (_i, [_vp, _vp, _vp, _i, _vp]),
    "reid_diou_cost": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "reid_rank_eval": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "reid_rank_eval_dev": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "reid_comm_unique_id": (_i, [_vp]),
    "reid_comm_init": (_i, [_vp, _i, _i, _vp]),
    "reid_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "reid_comm_destroy": (_i, [_vp]),
    "reid_allgather_dev": (_i, [_vp, _vp, _vp, _sz]),
    "reid_allgather_rows_dev": (_i, [_vp, _vp, _i, _sz, _vp, _vp, C.POINTER(_i)]),
    "reid_allreduce_f64": (_i, [_vp, C.POINTER(C.c_double), _i, _i]),
    "reid_knn_gallery_sharded_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "reid_conv2d_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "reid_gemm_nt": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
}
EXPORTS = tuple(sorted(_SIGS))

DEBUG_LIB_PATH = os.path.join(_HERE, "libreid_hip_debug.so")
DEBUG_EXPORTS = ("reid_debug_coissue", "reid_debug_comm_loopback", "reid_debug_conv_c64", "reid_debug_conv_split", "reid_debug_conv_diag", "reid_debug_conv_f16", "reid_debug_conv_f32",
                 "reid_debug_feed", "reid_debug_gemm_f16", "reid_debug_knn_merge", "reid_debug_knn_wide", "reid_debug_linear", "reid_debug_linear_rows", "reid_debug_mfma_bare", "reid_debug_mfma_shape", "reid_debug_select_exp", "reid_debug_set_switch", "reid_debug_get_switch", "reid_debug_two_linear", "reid_debug_two_linear_ablate")   # include/reid_hip_debug.h

_lib = None
_dbg = None


class ReidHipError(RuntimeError):
    """A non-zero reid_status; ``status`` carries it (include/reid_hip.h: REID_ERR_ARG -1, _HIP -2, _STATE -3, _NOMEM -4)."""
    status = None


def lib():
    """Loads libreid_hip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ReidHipError(
                "HIP extension missing: %s (build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C real-time-reid-tracking_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def debug_lib():
    """libreid_hip_debug.so (experiments + harnesses, include/reid_hip_debug.h); loads the product library first."""
    global _dbg
    if _dbg is None:
        lib()
        if not os.path.exists(DEBUG_LIB_PATH):
            raise ReidHipError("debug library missing: %s (make -C real-time-reid-tracking_amd/csrc)" % DEBUG_LIB_PATH)
        _dbg = C.CDLL(DEBUG_LIB_PATH)
    return _dbg


def check(status):
    if status != 0:
        msg = lib().reid_last_error()
        err = ReidHipError("libreid_hip status %d: %s" % (status, msg.decode() if msg else "?"))
        err.status = int(status)
        raise err
