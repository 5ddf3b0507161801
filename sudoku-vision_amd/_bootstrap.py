"""Lets cv/*.py and ml/model.py be imported the way the reference's callers import them -- as
top-level modules found through sys.path (pipeline/run.py:28-35: `from preprocess import ...`) --
as well as as package modules (`sudoku_vision_amd.cv.preprocess`)."""
import os
import sys


def package():
    try:
        import sudoku_vision_amd
    except ImportError:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import sudoku_vision_amd
    return sudoku_vision_amd
