"""MI355X drop-in for the reference's cv/extract.py (same names, arguments, return types)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import package  # noqa: E402
sys.path.pop(0)
_rt = package().runtime


def _to_dev(a, ctx):
    if isinstance(a, torch.Tensor):
        return a.contiguous(), True
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError(f"expected uint8 image, got {a.dtype}")
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device), False


def extract_cells(grid_image, cell_size: int = 28, margin_ratio: float = 0.1):
    """81 cell images, row-major, from a warped grid (reference cv/extract.py:13-56).

    Returns a Python list of 81 separate (cell_size, cell_size) uint8 arrays (tensors for tensor input)."""
    ctx = _rt.default_context()
    d, was = _to_dev(grid_image, ctx)
    h, w = d.shape[0], d.shape[1]
    margin_h, margin_w = int((h // 9) * margin_ratio), int((w // 9) * margin_ratio)
    cells = ctx.extract_cells(d, cell_size, margin_h, margin_w)
    if was:
        return [cells[i] for i in range(81)]
    host = cells.cpu().numpy()
    return [host[i].copy() for i in range(81)]


def is_cell_empty(cell, threshold: float = 0.02) -> bool:
    """True if the Otsu-binarised ink share of the cell is below `threshold` (reference cv/extract.py:59-79)."""
    ctx = _rt.default_context()
    d, _ = _to_dev(cell, ctx)
    if d.dim() != 2:
        raise ValueError("is_cell_empty expects a grayscale cell")
    ratio, _ = ctx.cell_ink_ratio(d[None])
    return bool(float(ratio[0]) < threshold)


def preprocess_cell_for_model(cell):
    """(1,28,28) float32 in [0,1] (reference cv/extract.py:82-99)."""
    cell = np.asarray(cell)
    if len(cell.shape) == 3:
        cell = package().cv.preprocess.grayscale(cell)
    if cell.shape != (28, 28):
        ctx = _rt.default_context()
        cell = ctx.resize_linear(_to_dev(cell, ctx)[0], (28, 28)).cpu().numpy()
    return (cell.astype(np.float32) / 255.0).reshape(1, 28, 28)
