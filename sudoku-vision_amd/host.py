"""Host (CPU) half of the hot path: the contour-based grid corner search of cv/grid.py:16-71, implemented
in C++ inside libsudokuvision_hip.so (csrc/host_contours.cpp).  Needs no GPU."""
import ctypes as C
import os

import numpy as np

from . import _native


def _bin(binary):
    b = np.asarray(binary)
    if b.dtype != np.uint8 or b.ndim != 2:
        raise TypeError("expected a 2-D uint8 binary image")
    return np.ascontiguousarray(b)


def find_contours(binary):
    """-> list of int32 arrays of shape (N,1,2), in cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) order."""
    b = _bin(binary)
    H, W = b.shape
    lib = _native.lib()
    npts, ncont = C.c_long(), C.c_int()
    rc = lib.sv_find_contours_u8(b.ctypes.data_as(C.c_void_p), H, W, W, None, 0, None, 0, C.byref(npts), C.byref(ncont))
    if rc not in (0, -6):
        _native.check(rc, "sv_find_contours_u8")
    pts = np.empty((max(npts.value, 1), 2), np.int32)
    sizes = np.empty(max(ncont.value, 1), np.int32)
    _native.check(lib.sv_find_contours_u8(b.ctypes.data_as(C.c_void_p), H, W, W, pts.ctypes.data_as(C.c_void_p), npts.value,
                                          sizes.ctypes.data_as(C.c_void_p), ncont.value, C.byref(npts), C.byref(ncont)), "sv_find_contours_u8")
    out, o = [], 0
    for i in range(ncont.value):
        out.append(pts[o:o + sizes[i]].reshape(-1, 1, 2).copy())
        o += sizes[i]
    return out


def find_contours_bits(bits, H, W):
    """bits uint32/int32 [H, W//32] (1 bit per pixel, LSB = leftmost) -> the same list find_contours gives for the unpacked image."""
    b = np.ascontiguousarray(bits).view(np.uint32)
    lib = _native.lib()
    npts, ncont = C.c_long(), C.c_int()
    rc = lib.sv_find_contours_bits(b.ctypes.data_as(C.c_void_p), int(H), int(W), None, 0, None, 0, C.byref(npts), C.byref(ncont))
    if rc not in (0, -6):
        _native.check(rc, "sv_find_contours_bits")
    pts = np.empty((max(npts.value, 1), 2), np.int32)
    sizes = np.empty(max(ncont.value, 1), np.int32)
    _native.check(lib.sv_find_contours_bits(b.ctypes.data_as(C.c_void_p), int(H), int(W), pts.ctypes.data_as(C.c_void_p), npts.value,
                                            sizes.ctypes.data_as(C.c_void_p), ncont.value, C.byref(npts), C.byref(ncont)), "sv_find_contours_bits")
    out, o = [], 0
    for i in range(ncont.value):
        out.append(pts[o:o + sizes[i]].reshape(-1, 1, 2).copy())
        o += sizes[i]
    return out


def _pts(contour):
    return np.ascontiguousarray(np.asarray(contour).reshape(-1, 2), np.int32)


def contour_area(contour) -> float:
    c = _pts(contour)
    v = C.c_double()
    _native.check(_native.lib().sv_contour_area_i32(c.ctypes.data_as(C.c_void_p), c.shape[0], C.byref(v)), "sv_contour_area_i32")
    return v.value


def arc_length(contour, closed=True) -> float:
    c = _pts(contour)
    v = C.c_double()
    _native.check(_native.lib().sv_arc_length_i32(c.ctypes.data_as(C.c_void_p), c.shape[0], int(bool(closed)), C.byref(v)), "sv_arc_length_i32")
    return v.value


def approx_poly_dp(contour, epsilon, closed=True):
    c = _pts(contour)
    out = np.empty((max(c.shape[0], 1), 2), np.int32)
    n = C.c_int()
    _native.check(_native.lib().sv_approx_poly_dp_i32(c.ctypes.data_as(C.c_void_p), c.shape[0], float(epsilon), int(bool(closed)),
                                                      out.ctypes.data_as(C.c_void_p), C.byref(n)), "sv_approx_poly_dp_i32")
    return out[:n.value].reshape(-1, 1, 2).copy()


def find_grid_corners(binary, min_area_ratio=0.1, epsilon_ratio=0.02):
    """-> int32 (4,2) or None."""
    b = _bin(binary)
    H, W = b.shape
    corners = np.empty((4, 2), np.int32)
    rc = _native.lib().sv_find_grid_corners_u8(b.ctypes.data_as(C.c_void_p), H, W, W, float(min_area_ratio), float(epsilon_ratio),
                                               corners.ctypes.data_as(C.c_void_p))
    if rc < 0:
        _native.check(rc, "sv_find_grid_corners_u8")
    return corners if rc == 1 else None


def find_grid_corners_batch(binaries, min_area_ratio=0.1, epsilon_ratio=0.02, threads=None):
    """binaries uint8 [n,H,W] (host) -> (corners int32 [n,4,2], found bool [n])."""
    b = np.asarray(binaries)
    if b.dtype != np.uint8 or b.ndim != 3:
        raise TypeError("expected uint8 [n,H,W]")
    b = np.ascontiguousarray(b)
    n, H, W = b.shape
    corners = np.zeros((n, 4, 2), np.int32)
    found = np.zeros(n, np.uint8)
    threads = threads or min(n, os.cpu_count() or 1)
    _native.check(_native.lib().sv_find_grid_corners_batch_u8(b.ctypes.data_as(C.c_void_p), n, H, W, W, H * W, float(min_area_ratio),
                                                              float(epsilon_ratio), corners.ctypes.data_as(C.c_void_p),
                                                              found.ctypes.data_as(C.c_void_p), int(threads)), "sv_find_grid_corners_batch_u8")
    return corners, found.astype(bool)


def find_grid_corners_bits_batch(bits, H, W, min_area_ratio=0.1, epsilon_ratio=0.02, threads=None):
    """bits int32/uint32 [n,H,W//32] (host; 1 bit per pixel, LSB = leftmost) -> (corners int32 [n,4,2], found bool [n])."""
    b = np.ascontiguousarray(bits)
    n = b.shape[0]
    corners = np.zeros((n, 4, 2), np.int32)
    found = np.zeros(n, np.uint8)
    threads = threads or min(n, os.cpu_count() or 1)
    _native.check(_native.lib().sv_find_grid_corners_bits_batch(b.ctypes.data_as(C.c_void_p), n, int(H), int(W), float(min_area_ratio), float(epsilon_ratio),
                                                                corners.ctypes.data_as(C.c_void_p), found.ctypes.data_as(C.c_void_p), int(threads)),
                  "sv_find_grid_corners_bits_batch")
    return corners, found.astype(bool)


def set_pool_affinity(cpus):
    """Restrict the library's host worker threads to these CPUs (an iterable of ints; empty = no restriction for new workers)."""
    a = np.ascontiguousarray(sorted(int(c) for c in cpus), np.int32)
    _native.check(_native.lib().sv_host_pool_set_affinity(a.ctypes.data_as(C.c_void_p) if a.size else None, int(a.size)), "sv_host_pool_set_affinity")


def sparse_bits_record_bytes(H, W, cap_values):
    """Bytes of one sparse record (sv_pack_sparse_bits) with room for cap_values non-zero words."""
    r = _native.lib().sv_sparse_bits_record_bytes(int(H), int(W), int(cap_values))
    if r < 0:
        raise ValueError("bad shape for a sparse record (W must be a multiple of 32)")
    return int(r)


def find_grid_corners_sparse_batch(records, H, W, min_area_ratio=0.1, epsilon_ratio=0.02, threads=None):
    """records uint8 [n,stride] (host; Context.pack_sparse_bits' output after the D2H copy) -> (corners int32 [n,4,2], found uint8 [n]);
    found[i] == 2: record i overflowed its capacity and has to be searched from the dense bit image."""
    if records.dtype != np.uint8 or records.ndim != 2 or not records.flags.c_contiguous:
        raise TypeError("records must be a C-contiguous uint8 [n,stride] array")
    n, stride = records.shape
    corners = np.zeros((n, 4, 2), np.int32)
    found = np.zeros(n, np.uint8)
    threads = threads or min(n, os.cpu_count() or 1)
    _native.check(_native.lib().sv_find_grid_corners_sparse_batch(records.ctypes.data_as(C.c_void_p), stride, n, int(H), int(W), float(min_area_ratio),
                                                                  float(epsilon_ratio), corners.ctypes.data_as(C.c_void_p),
                                                                  found.ctypes.data_as(C.c_void_p), int(threads)), "sv_find_grid_corners_sparse_batch")
    return corners, found


def sparse_bits_expand(record, H, W):
    """One sparse record (uint8 [stride], host) -> the dense bit image uint32 [H, W//32]."""
    rec = np.ascontiguousarray(record, np.uint8)
    out = np.empty((H, W // 32), np.uint32)
    _native.check(_native.lib().sv_sparse_bits_expand(rec.ctypes.data_as(C.c_void_p), int(H), int(W), out.ctypes.data_as(C.c_void_p)), "sv_sparse_bits_expand")
    return out


def solve_sudoku(grid):
    """grid: 9x9 (or 81) digits, 0 = empty -> (code, solution 9x9 uint8); code 1 solved, 0 no solution, -1 invalid input
    (the reference solver's SOLVE_* codes).  solution == grid unless solved."""
    g = np.ascontiguousarray(np.asarray(grid).reshape(81))
    if g.min() < 0 or g.max() > 255:
        return -1, np.asarray(grid, np.uint8).reshape(9, 9)
    g = g.astype(np.uint8)
    out = np.empty(81, np.uint8)
    res = C.c_int()
    _native.check(_native.lib().sv_solve_sudoku(g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(res)), "sv_solve_sudoku")
    return res.value, out.reshape(9, 9)


# ---- JPEG front end, host half (csrc/host_jpeg.cpp): header parsing and Huffman decoding -------------------
def jpeg_parse(data: bytes):
    """-> _native.JpegInfo (sv_jpeg_info).  Raises NativeError for non-JPEG data and for unsupported JPEG flavours."""
    info = _native.JpegInfo()
    _native.check(_native.lib().sv_jpeg_parse(data, len(data), C.byref(info)), "sv_jpeg_parse")
    return info


def jpeg_entropy_decode(data: bytes, coef=None, quant=None, threads=1):
    """-> (info, coef int16 [coef_count], quant uint16 [3,64]).  coef/quant may be caller-owned (e.g. pinned) numpy arrays."""
    info = jpeg_parse(data)
    if coef is None:
        coef = np.empty(info.coef_count, np.int16)
    if quant is None:
        quant = np.empty((3, 64), np.uint16)
    if coef.dtype != np.int16 or coef.size < info.coef_count or not coef.flags.c_contiguous:
        raise ValueError("coef must be a contiguous int16 array of at least info.coef_count values")
    _native.check(_native.lib().sv_jpeg_entropy_decode(data, len(data), coef.ctypes.data_as(C.c_void_p), quant.ctypes.data_as(C.c_void_p), int(threads)),
                  "sv_jpeg_entropy_decode")
    return info, coef, quant


def jpeg_entropy_decode_sparse(data: bytes, threads=1):
    """-> (info, masks uint64 [blocks], offsets uint32 [blocks], values int16 [used], quant uint16 [3,64]): the compact
    transport form (mask over zigzag positions + first-value index per block, non-zero values in zigzag order)."""
    info = jpeg_parse(data)
    nb = info.coef_count // 64
    masks, offs = np.empty(nb, np.uint64), np.empty(nb, np.uint32)
    vals = np.empty(info.sparse_capacity, np.int16)
    quant = np.empty((3, 64), np.uint16)
    used = C.c_long()
    _native.check(_native.lib().sv_jpeg_entropy_decode_sparse(data, len(data), masks.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                                                              vals.ctypes.data_as(C.c_void_p), vals.size, C.byref(used), quant.ctypes.data_as(C.c_void_p), int(threads)),
                  "sv_jpeg_entropy_decode_sparse")
    return info, masks, offs, vals[:used.value], quant
