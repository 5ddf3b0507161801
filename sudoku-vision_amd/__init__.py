"""sudoku-vision frame -> digits hot path on MI355X (gfx950).

Host-side mirror of the reference's interface for this path -- cv.preprocess / cv.grid / cv.extract /
ml.model keep the reference's names, arguments and error behaviour -- over the C ABI of
csrc/libsudokuvision_hip.so (include/sudoku_vision_hip.h).  There is no CPU fallback: every function
that computes needs the HIP library and a GPU, and says so when either is missing.
"""
from . import _native  # noqa: F401  (does not load the library until first use)
from . import host  # noqa: F401
from .runtime import Context, default_context, frames_to_digits  # noqa: F401

__version__ = "0.1.0"
