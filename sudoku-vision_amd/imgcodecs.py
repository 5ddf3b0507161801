"""MI355X counterpart of the one cv2.imgcodecs call on the path: `cv2.imread(path)` (pipeline/run.py:250,
pipeline/run_v2.py:267, tests/test_integration.py:126) for JPEG files -- same return convention: a BGR uint8 numpy array,
or None when the file cannot be read or is not a decodable image (cv2.imread does not raise).  JPEG flavours this build
does not decode (progressive, arithmetic, CMYK, 12-bit) raise NativeError instead of returning a wrong image.
"""
import os

from . import _native
from .runtime import default_context


def imdecode(buf, device=False, ctx=None, threads=1):
    """bytes of a JPEG file -> BGR image; device=True keeps it on the GPU (CUDA uint8 tensor) for K1/K2."""
    data = bytes(buf)
    ctx = ctx or default_context()
    try:
        img = ctx.imdecode(data, threads=threads)
    except _native.NativeError as e:
        if "SV_ERR_UNSUPPORTED" in str(e):
            raise
        return None
    return img if device else img.cpu().numpy()


def imread(path, device=False, ctx=None, threads=1):
    try:
        with open(os.fspath(path), "rb") as f:
            data = f.read()
    except OSError:
        return None
    return imdecode(data, device=device, ctx=ctx, threads=threads)
