"""ctypes binding of libsudokuvision_hip.so (include/sudoku_vision_hip.h).  Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsudokuvision_hip.so")
# test-only superset build (csrc/Makefile, include/sudoku_vision_xcheck.h): independent second implementations for cross-checks.  Loaded by
# tests/ and tools/ through lib_xcheck(); never by the package itself.
XCHECK_LIB_PATH = os.path.join(_HERE, "csrc", "libsudokuvision_xcheck.so")
ABI_VERSION = 2

SV_OK = 0
ERR_NAMES = {-1: "SV_ERR_BAD_ARG", -2: "SV_ERR_HIP", -3: "SV_ERR_NO_WEIGHTS", -4: "SV_ERR_UNSUPPORTED",
             -5: "SV_ERR_DEGENERATE", -6: "SV_ERR_BUFFER"}

_p, _i, _l, _d, _f, _pd = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_float, C.c_ssize_t

# name -> argtypes; every function returns int unless listed in _RESTYPES
SIGNATURES = {
    "sv_version": [],
    "sv_last_error": [],
    "sv_ctx_create": [_i, C.POINTER(_p)],
    "sv_ctx_destroy": [_p],
    "sv_ctx_reserve": [_p, _l],
    "sv_ctx_set_precision": [_p, _i],
    "sv_load_weights_f32": [_p, _p],
    "sv_timing_begin": [_p],
    "sv_timing_end": [_p, _p, _p, _i],
    "sv_ctx_set_cnn_kernels": [_p, _i],
    "sv_conv_kernel_info": [_p, _p, _p, _p, _p, _p],
    "sv_gray_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p],
    "sv_blur_u8": [_p, _p, _i, _i, _i, _i, _p, _p],
    "sv_adaptive_threshold_u8": [_p, _p, _i, _i, _i, _i, _d, _i, _p, _p],
    "sv_preprocess_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p],
    "sv_solve_sudoku": [_p, _p, _p],
    "sv_despeckle_u8": [_p, _p, _i, _i, _i, _p, _p, _p],
    "sv_preprocess_bits_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p],
    "sv_preprocess_warp_cells_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p, _p, _p],
    "sv_despeckle_bits": [_p, _p, _i, _i, _i, _p],
    "sv_copy_to_pinned_host": [_p, _p, _p, C.c_size_t, _p],
    "sv_sparse_bits_record_bytes": [_i, _i, _l],
    "sv_pack_sparse_bits": [_p, _p, _i, _i, _i, _p, _l, _p],
    "sv_find_grid_corners_sparse_batch": [_p, _l, _i, _i, _i, _d, _d, _p, _p, _i],
    "sv_sparse_bits_expand": [_p, _i, _i, _p],
    "sv_host_pool_set_affinity": [_p, _i],
    "sv_find_grid_corners_bits_batch": [_p, _i, _i, _i, _d, _d, _p, _p, _i],
    "sv_corners_to_minv": [_p, _i, _i, _f, _p],
    "sv_corners_to_minv_batch": [_p, _i, _i, _f, _p, _p],
    "sv_find_grid_corners_u8": [_p, _i, _i, _pd, _d, _d, _p],
    "sv_find_grid_corners_batch_u8": [_p, _i, _i, _i, _pd, _pd, _d, _d, _p, _p, _i],
    "sv_find_contours_u8": [_p, _i, _i, _pd, _p, _l, _p, _i, _p, _p],
    "sv_find_contours_bits": [_p, _i, _i, _p, _l, _p, _i, _p, _p],
    "sv_contour_area_i32": [_p, _i, _p],
    "sv_arc_length_i32": [_p, _i, _i, _p],
    "sv_approx_poly_dp_i32": [_p, _i, _d, _i, _p, _p],
    "sv_warp_perspective_u8": [_p, _p, _i, _i, _pd, _i, _p, _i, _p, _p],
    "sv_extract_cells_u8": [_p, _p, _i, _i, _pd, _i, _i, _i, _i, _p, _p],
    "sv_warp_cells_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p, _p],
    "sv_resize_linear_u8": [_p, _p, _i, _i, _pd, _p, _i, _i, _p],
    "sv_cell_ink_ratio_u8": [_p, _p, _l, _i, _p, _p, _p],
    "sv_preprocess_cells_u8": [_p, _p, _l, _p, _p],
    "sv_jpeg_parse": [_p, C.c_size_t, _p],
    "sv_jpeg_entropy_decode": [_p, C.c_size_t, _p, _p, _i],
    "sv_jpeg_entropy_decode_sparse": [_p, C.c_size_t, _p, _p, _p, _l, _p, _p, _i],
    "sv_jpeg_entropy_decode_batch": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "sv_jpeg_reconstruct_bgr_u8": [_p, _p, _p, _p, _p, _pd, _p],
    "sv_jpeg_reconstruct_sparse_bgr_u8": [_p, _p, _p, _p, _p, _p, _p, _pd, _p],
    "sv_softmax_topk_f32": [_p, _p, _l, _i, _p, _p, _p],
    "sv_cnn_forward_f32": [_p, _p, _l, _p, _p, _p, _p],
    "sv_cnn_forward_cells_u8": [_p, _p, _l, _i, _p, _p, _p, _p],
    "sv_frames_to_digits": [_p, _p, _i, _i, _i, _pd, _pd, _p, _i, _p, _p, _p, _p, _p],
}
_RESTYPES = {"sv_last_error": C.c_char_p, "sv_sparse_bits_record_bytes": C.c_long}
# what libsudokuvision_xcheck.so exports on top of SIGNATURES (include/sudoku_vision_xcheck.h)
XCHECK_SIGNATURES = {
    "sv_preprocess_stats": [_p, _p, _p],
    "sv_preprocess_mm_u8": [_p, _p, _i, _i, _i, _pd, _pd, _p, _p, _p],
    "svx_ctx_set_fc_frame_kernel": [_p, _i],
}



class JpegInfo(C.Structure):
    """sv_jpeg_info of include/sudoku_vision_hip.h"""
    _fields_ = [(n, C.c_int) for n in ("width", "height", "out_width", "out_height", "components", "h_samp", "v_samp",
                                        "orientation", "restart_interval")] + [("coef_count", C.c_long), ("sparse_capacity", C.c_long)]


_lib = None
_xlib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Loads the HIP library once.  torch is imported first so that the process holds a single HIP
    runtime (torch's libamdhip64 and the one this library links share a soname)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C sudoku-vision_amd/csrc`.  There is no CPU fallback.")
        _lib = _bind(C.CDLL(LIB_PATH), SIGNATURES)
    return _lib


def _bind(handle, signatures):
    for name, args in signatures.items():
        fn = getattr(handle, name)  # AttributeError = ABI mismatch, let it surface
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = handle.sv_version()
    if v != ABI_VERSION:
        raise NativeError(f"{getattr(handle, '_name', 'library')}: ABI version {v}, this package binds version {ABI_VERSION}: rebuild (make -C sudoku-vision_amd/csrc)")
    return handle


def lib_xcheck():
    """The test-only superset library (cross-check kernels).  For tests/ and tools/ only."""
    global _xlib
    if _xlib is None:
        import torch  # noqa: F401
        if not os.path.exists(XCHECK_LIB_PATH):
            raise NativeError(f"{XCHECK_LIB_PATH} is missing: make -C sudoku-vision_amd/csrc libsudokuvision_xcheck.so")
        _xlib = _bind(C.CDLL(XCHECK_LIB_PATH), {**SIGNATURES, **XCHECK_SIGNATURES})
    return _xlib


def check(rc: int, what: str = "", handle=None):
    if rc != SV_OK:
        msg = (handle or lib()).sv_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what or 'libsudokuvision_hip'}: {ERR_NAMES.get(rc, rc)}: {msg}")
