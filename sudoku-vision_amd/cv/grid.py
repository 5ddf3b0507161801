"""MI355X drop-in for the reference's cv/grid.py (same names, arguments, return types).

The perspective warp runs on the GPU (csrc/k2_warp_cells.hip); the homography is solved on the host in
fp64 (csrc/sv_api.cpp).  The contour-based corner search stays on the host CPU, as in the reference.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import package  # noqa: E402
sys.path.pop(0)
_pkg = package()
_rt = _pkg.runtime
_host = _pkg.host


def find_contours(binary):
    """Find all (external) contours in a binary image (reference cv/grid.py:16-21)."""
    return _host.find_contours(binary)


def approximate_polygon(contour, epsilon_ratio: float = 0.02):
    """Approximate contour with a polygon at epsilon_ratio * perimeter (reference cv/grid.py:24-34)."""
    perimeter = _host.arc_length(contour, closed=True)
    return _host.approx_poly_dp(contour, epsilon_ratio * perimeter, closed=True)


def find_grid_contour(binary, min_area_ratio: float = 0.1):
    """The sudoku grid contour = largest quadrilateral; 4 corner points or None (reference cv/grid.py:37-71)."""
    if isinstance(binary, torch.Tensor):
        binary = binary.cpu().numpy()
    return _host.find_grid_corners(binary, min_area_ratio=min_area_ratio, epsilon_ratio=0.02)


def order_points(pts):
    """Order 4 points as top-left, top-right, bottom-right, bottom-left (reference cv/grid.py:74-91)."""
    pts = np.asarray(pts)
    rect = np.zeros((4, 2), dtype=np.float32)
    s = pts.sum(axis=1)
    rect[0] = pts[np.argmin(s)]
    rect[2] = pts[np.argmax(s)]
    d = np.diff(pts, axis=1)
    rect[1] = pts[np.argmin(d)]
    rect[3] = pts[np.argmax(d)]
    return rect


def warp_perspective(image, corners, output_size: int = 450, inset_ratio: float = 0.0):
    """Warp the grid region to an output_size square (reference cv/grid.py:94-133)."""
    ctx = _rt.default_context()
    was_tensor = isinstance(image, torch.Tensor)
    if was_tensor:
        d = image.contiguous()
    else:
        image = np.asarray(image)
        if image.dtype != np.uint8:
            raise TypeError(f"expected uint8 image, got {image.dtype}")
        d = torch.from_numpy(np.ascontiguousarray(image)).to(ctx.device)
    corners = np.asarray(corners.cpu() if isinstance(corners, torch.Tensor) else corners).astype(np.float32).reshape(1, 4, 2)
    minv = ctx.minv_to_device(_rt.Context.corners_to_minv(corners, output_size, inset_ratio))
    out = ctx.warp_perspective(d, minv, output_size)
    return out if was_tensor else out.cpu().numpy()
