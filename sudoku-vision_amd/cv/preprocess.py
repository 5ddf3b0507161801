"""MI355X drop-in for the reference's cv/preprocess.py (same names, arguments, return types).

numpy uint8 in -> numpy uint8 out (what pipeline/run.py expects); a CUDA uint8 tensor in -> a CUDA
tensor out with no host round trip.  All arithmetic runs in the HIP kernels of csrc/k1_threshold.hip.
"""
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
        if a.dtype != torch.uint8 or not a.is_cuda:
            raise TypeError("expected a uint8 CUDA tensor or a numpy uint8 array")
        return a.contiguous(), True
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError(f"expected uint8 image, got {a.dtype}")
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device), False


def _back(t, was_tensor):
    return t if was_tensor else t.cpu().numpy()


def grayscale(image):
    """Convert BGR image to grayscale (reference cv/preprocess.py:15-19)."""
    if len(image.shape) == 2:
        return image  # already grayscale: the reference returns its argument
    ctx = _rt.default_context()
    d, was = _to_dev(image, ctx)
    return _back(ctx.gray(d[None])[0], was)


def blur(image, ksize: int = 5):
    """Gaussian blur, sigma derived from ksize (reference cv/preprocess.py:22-29)."""
    ctx = _rt.default_context()
    d, was = _to_dev(image, ctx)
    if d.dim() != 2:
        raise ValueError("blur expects a single-channel image")
    return _back(ctx.blur(d[None], ksize)[0], was)


def threshold(image, block_size: int = 11, c: int = 2):
    """Adaptive Gaussian threshold, inverted binary (reference cv/preprocess.py:32-54)."""
    ctx = _rt.default_context()
    d, was = _to_dev(image, ctx)
    if d.dim() != 2:
        raise ValueError("threshold expects a single-channel image")
    return _back(ctx.adaptive_threshold(d[None], block_size, c, inv=True)[0], was)


def preprocess_for_grid_detection(image):
    """grayscale -> blur(5) -> threshold(11, 2) in one fused kernel (reference cv/preprocess.py:57-65)."""
    ctx = _rt.default_context()
    d, was = _to_dev(image, ctx)
    if d.dim() == 2:
        return _back(ctx.adaptive_threshold(ctx.blur(d[None], 5), 11, 2, inv=True)[0], was)
    return _back(ctx.preprocess(d[None])[0], was)
