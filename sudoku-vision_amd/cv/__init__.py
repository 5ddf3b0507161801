"""Computer-vision half of the hot path; same exports as the reference's cv/__init__.py:8-19."""
from .preprocess import grayscale, threshold, blur
from .grid import find_grid_contour, warp_perspective
from .extract import extract_cells

__all__ = ["grayscale", "threshold", "blur", "find_grid_contour", "warp_perspective", "extract_cells"]
