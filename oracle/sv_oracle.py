"""ctypes + numpy front end of the CPU oracle (oracle/sv_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (sudoku-vision_amd/) never does.  Parity status of what it wraps: see the header of
sv_oracle.c ("parity unpinned" for the OpenCV-defined stages).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsv_oracle.so")


def build(force: bool = False) -> str:
    """Compile sv_oracle.c with gcc (make -C oracle) if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("sv_oracle.c", "sv_jpeg_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def gray(bgr):
    bgr, p = _u8(bgr)
    H, W = bgr.shape[:2]
    out = np.empty((H, W), np.uint8)
    lib().svo_gray_bgr(p, H, W, C.c_long(W * 3), out.ctypes.data_as(C.c_void_p))
    return out


def gaussian_blur(img, ksize=5):
    img, p = _u8(img)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    if lib().svo_gaussian_blur_u8(p, H, W, int(ksize), out.ctypes.data_as(C.c_void_p)) != 0:
        raise NotImplementedError(f"oracle: GaussianBlur ksize={ksize} not restated")
    return out


def gaussian_kernel_f32(n):
    out = np.empty(n, np.float32)
    if lib().svo_gaussian_kernel_f32(int(n), out.ctypes.data_as(C.c_void_p)) != 0:
        raise ValueError(n)
    return out


def adaptive_mean(img, block=11):
    img, p = _u8(img)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    if lib().svo_adaptive_mean_u8(p, H, W, int(block), out.ctypes.data_as(C.c_void_p)) != 0:
        raise ValueError(block)
    return out


def adaptive_threshold(img, block=11, c=2, inv=True):
    img, p = _u8(img)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    rc = lib().svo_adaptive_threshold_u8(p, H, W, int(block), C.c_double(float(c)), int(bool(inv)),
                                         out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(block)
    return out


def preprocess_for_grid_detection(bgr):
    bgr, p = _u8(bgr)
    H, W = bgr.shape[:2]
    out = np.empty((H, W), np.uint8)
    rc = lib().svo_preprocess_for_grid_detection(p, H, W, C.c_long(W * 3), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def order_points(pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(8)
    out = np.empty(8, np.float32)
    lib().svo_order_points(pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out.reshape(4, 2)


def get_perspective_transform(src, dst):
    src = np.ascontiguousarray(src, np.float32).reshape(8)
    dst = np.ascontiguousarray(dst, np.float32).reshape(8)
    M = np.empty(9, np.float64)
    rc = lib().svo_get_perspective_transform(src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                             M.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("singular")
    return M.reshape(3, 3)


def invert3x3(M):
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    D = np.empty(9, np.float64)
    if lib().svo_invert3x3(M.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p)) != 0:
        raise ValueError("singular")
    return D.reshape(3, 3)


def corners_to_minv(corners, out_size=450, inset_ratio=0.0):
    c = np.ascontiguousarray(corners, np.float32).reshape(8)
    Minv = np.empty(9, np.float64)
    rc = lib().svo_corners_to_minv(c.ctypes.data_as(C.c_void_p), int(out_size), C.c_float(inset_ratio),
                                   Minv.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("degenerate corners")
    return Minv.reshape(3, 3)


def warp_perspective(img, corners, output_size=450, inset_ratio=0.0):
    img, p = _u8(img)
    H, W = img.shape[:2]
    Cn = 1 if img.ndim == 2 else img.shape[2]
    Minv = corners_to_minv(corners, output_size, inset_ratio).reshape(9)
    shape = (output_size, output_size) if img.ndim == 2 else (output_size, output_size, Cn)
    out = np.empty(shape, np.uint8)
    lib().svo_warp_perspective_u8(p, H, W, C.c_long(W * Cn), Cn, Minv.ctypes.data_as(C.c_void_p),
                                  int(output_size), out.ctypes.data_as(C.c_void_p))
    return out


def resize_linear(img, dsize):
    img, p = _u8(img)
    sh, sw = img.shape
    dw, dh = dsize
    out = np.empty((dh, dw), np.uint8)
    lib().svo_resize_linear_u8(p, sh, sw, C.c_long(sw), out.ctypes.data_as(C.c_void_p), dh, dw)
    return out


def extract_cells(grid, cell_size=28, margin_ratio=0.1):
    grid, p = _u8(grid)
    h, w = grid.shape[:2]
    Cn = 1 if grid.ndim == 2 else grid.shape[2]
    mh, mw = int((h // 9) * margin_ratio), int((w // 9) * margin_ratio)
    out = np.empty((81, cell_size, cell_size), np.uint8)
    lib().svo_extract_cells(p, h, w, C.c_long(w * Cn), Cn, int(cell_size), mh, mw, out.ctypes.data_as(C.c_void_p))
    return out


def warp_cells(bgr, corners, want_warped=False):
    bgr, p = _u8(bgr)
    H, W = bgr.shape[:2]
    c = np.ascontiguousarray(corners, np.float32).reshape(8)
    cells = np.empty((81, 28, 28), np.uint8)
    warped = np.empty((450, 450, 3), np.uint8) if want_warped else None
    rc = lib().svo_warp_cells(p, H, W, C.c_long(W * 3), c.ctypes.data_as(C.c_void_p),
                              cells.ctypes.data_as(C.c_void_p),
                              warped.ctypes.data_as(C.c_void_p) if want_warped else None)
    if rc != 0:
        raise ValueError("degenerate corners")
    return (cells, warped) if want_warped else cells


def cells_to_input(cells):
    cells, p = _u8(cells)
    x = np.empty(cells.shape, np.float32)
    lib().svo_cells_to_input_f32(p, C.c_long(cells.size), x.ctypes.data_as(C.c_void_p))
    return x


# ---- A5: host corner search (cv/grid.py:16-71) -------------------------------------------------------
def find_contours(binary):
    """-> list of int32 arrays (N,1,2), cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) order."""
    b, p = _u8(binary)
    H, W = b.shape
    pts, sizes, total = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.c_long()
    n = lib().svo_find_contours(p, H, W, C.c_long(W), C.byref(pts), C.byref(sizes), C.byref(total))
    out, o = [], 0
    allp = np.ctypeslib.as_array(pts, shape=(max(total.value, 1), 2)).copy()
    sz = np.ctypeslib.as_array(sizes, shape=(max(n, 1),)).copy()
    lib().svo_free(pts)
    lib().svo_free(sizes)
    for i in range(n):
        out.append(allp[o:o + sz[i]].reshape(-1, 1, 2).astype(np.int32))
        o += sz[i]
    return out


def _pts(c):
    c = np.ascontiguousarray(np.asarray(c).reshape(-1, 2), np.int32)
    return c, c.ctypes.data_as(C.c_void_p)


def contour_area(contour):
    c, p = _pts(contour)
    lib().svo_contour_area.restype = C.c_double
    return lib().svo_contour_area(p, c.shape[0])


def arc_length(contour, closed=True):
    c, p = _pts(contour)
    lib().svo_arc_length.restype = C.c_double
    return lib().svo_arc_length(p, c.shape[0], int(closed))


def approx_poly_dp(contour, epsilon, closed=True):
    c, p = _pts(contour)
    dst = np.empty((max(c.shape[0], 1), 2), np.int32)
    n = lib().svo_approx_poly_dp(p, c.shape[0], C.c_double(epsilon), int(closed), dst.ctypes.data_as(C.c_void_p))
    return dst[:n].reshape(-1, 1, 2).copy()


def find_grid_contour(binary, min_area_ratio=0.1, epsilon_ratio=0.02):
    b, p = _u8(binary)
    H, W = b.shape
    corners = np.empty((4, 2), np.int32)
    found = lib().svo_find_grid_contour(p, H, W, C.c_long(W), C.c_double(min_area_ratio), C.c_double(epsilon_ratio),
                                        corners.ctypes.data_as(C.c_void_p))
    return corners if found else None


# ---- N1: per-cell glue of pipeline/run.py:73-95 ------------------------------------------------------
def clahe(img, clip_limit=2.0, tiles=(4, 4)):
    img, p = _u8(img)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    if lib().svo_clahe_u8(p, H, W, C.c_double(clip_limit), int(tiles[0]), int(tiles[1]), out.ctypes.data_as(C.c_void_p)) != 0:
        raise NotImplementedError("oracle: CLAHE only for sizes divisible by the tile grid")
    return out


def preprocess_cells(cells):
    """run.py's preprocess_cell on u8 [n,28,28]: CLAHE(2.0,(4,4)) then adaptiveThreshold(GAUSSIAN, BINARY, 11, 2)."""
    cells, p = _u8(cells)
    out = np.empty(cells.shape, np.uint8)
    assert lib().svo_preprocess_cells(p, C.c_long(cells.size // 784), out.ctypes.data_as(C.c_void_p)) == 0
    return out


# ---- N3: is_cell_empty (cv/extract.py:59-79) -----------------------------------------------------------
def cell_ink_ratio(cell):
    """-> (non_zero / total of the Otsu BINARY_INV image, Otsu threshold)."""
    cell, p = _u8(cell)
    H, W = cell.shape
    t = C.c_int()
    lib().svo_cell_ink_ratio.restype = C.c_double
    return lib().svo_cell_ink_ratio(p, H, W, C.byref(t)), t.value


def is_cell_empty(cell, threshold=0.02):
    return cell_ink_ratio(cell)[0] < threshold


# ---- N4: cv2.imread on a baseline JPEG (oracle/sv_jpeg_oracle.c) ----
class JpegInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "out_width", "out_height", "components", "h_samp", "v_samp",
                                        "orientation", "restart_interval")] + [("coef_count", C.c_long)]


def jpeg_info(data: bytes):
    info = JpegInfo()
    rc = lib().svo_jpeg_info_parse(data, C.c_size_t(len(data)), C.byref(info))
    if rc:
        raise ValueError(f"oracle: jpeg header rc={rc}")
    return info


def jpeg_coefficients(data: bytes):
    """-> (coef int16 [coef_count] in the product's layout: per component, padded block grid row-major, 64 natural-order
    values per block; quant uint16 [components, 64] natural order)."""
    info = jpeg_info(data)
    coef = np.empty(info.coef_count, np.int16)
    quant = np.zeros((info.components, 64), np.uint16)
    rc = lib().svo_jpeg_coefficients(data, C.c_size_t(len(data)), coef.ctypes.data_as(C.c_void_p), quant.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"oracle: jpeg entropy decode rc={rc}")
    return coef, quant


def imdecode(data: bytes):
    """cv2.imread / cv2.imdecode(IMREAD_COLOR) of a baseline JPEG -> BGR uint8 [H,W,3] (EXIF orientation applied)."""
    info = jpeg_info(data)
    out = np.empty((info.out_height, info.out_width, 3), np.uint8)
    rc = lib().svo_jpeg_decode_bgr(data, C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"oracle: jpeg decode rc={rc}")
    return out
