"""CPU: the C-ABI library loads and exports every symbol include/sudoku_vision_hip.h declares; host-only
entry points (no GPU needed) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sv_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    import sudoku_vision_amd as sva
    if not os.path.exists(sva._native.LIB_PATH):
        g.build()
    return sva._native.lib()


def test_header_symbols_exported(lib):
    import sudoku_vision_amd as sva
    hdr = open(os.path.join(ROOT, "include", "sudoku_vision_hip.h")).read()
    declared = set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(sva._native.SIGNATURES), "ctypes table and header disagree"
    assert lib.sv_version() == sva._native.ABI_VERSION == 2


def test_xcheck_library_is_a_superset_and_the_product_is_lean(lib):
    """The test-only library exports everything the product does plus what include/sudoku_vision_xcheck.h declares; the product library
    exports none of those extras (cross-check kernels are not shipped)."""
    import sudoku_vision_amd as sva
    x = sva._native.lib_xcheck()
    hdr = open(os.path.join(ROOT, "include", "sudoku_vision_xcheck.h")).read()
    extra = set(re.findall(r"\b(svx?_[a-z0-9_]+)\s*\(", hdr))
    assert extra == set(sva._native.XCHECK_SIGNATURES)
    for name in extra:
        assert hasattr(x, name)
        assert not hasattr(lib, name), f"{name} leaked into the product library"
    for name in sva._native.SIGNATURES:
        assert hasattr(x, name)


def test_no_environment_switches_in_the_library():
    """The C ABI's behaviour depends on its arguments and setters only."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "sudoku-vision_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(f).read(), f


def test_corners_to_minv_matches_oracle(lib):
    import sudoku_vision_amd as sva
    rs = np.random.RandomState(0)
    base = np.array([[500, 100], [1400, 90], [1380, 980], [480, 1000]], np.float32)
    corners = np.stack([np.round(base + rs.uniform(-60, 60, (4, 2))).astype(np.float32)[rs.permutation(4)] for _ in range(16)])
    got = sva.Context.corners_to_minv(corners)
    for i in range(16):
        assert (got[i] == o.corners_to_minv(corners[i])).all()          # bit-exact fp64
    got = sva.Context.corners_to_minv(corners, 300, 0.05)
    for i in range(16):
        assert np.allclose(got[i], o.corners_to_minv(corners[i], 300, 0.05), rtol=1e-12, atol=0)


def test_degenerate_corners_error(lib):
    import sudoku_vision_amd as sva
    with pytest.raises(sva._native.NativeError, match="SV_ERR_DEGENERATE"):
        sva.Context.corners_to_minv(np.zeros((1, 4, 2), np.float32))


def test_bad_args_return_codes(lib):
    assert lib.sv_corners_to_minv(None, 1, 450, 0.0, None) == -1
    assert b"bad argument" in lib.sv_last_error()
    assert lib.sv_ctx_destroy(None) == 0


def test_no_gpu_fails_loudly():
    import torch
    import sudoku_vision_amd as sva
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sva._native.NativeError, match="no CPU fallback"):
        sva.default_context()
    from sudoku_vision_amd.cv.preprocess import preprocess_for_grid_detection
    with pytest.raises(sva._native.NativeError):
        preprocess_for_grid_detection(np.zeros((32, 32, 3), np.uint8))


def test_dropin_module_names_resolve():
    """The reference's callers do sys.path.insert(cv/), `from preprocess import ...` (pipeline/run.py:28-35)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from preprocess import preprocess_for_grid_detection, grayscale, blur, threshold;"
            "from grid import find_grid_contour, warp_perspective, order_points, find_contours, approximate_polygon;"
            "from extract import extract_cells, is_cell_empty, preprocess_cell_for_model;"
            "from model import DigitCNN, count_parameters; m = DigitCNN();"
            "assert count_parameters(m) == 421642;"
            "assert list(m.state_dict()) == ['conv1.weight','conv1.bias','conv2.weight','conv2.bias','fc1.weight','fc1.bias','fc2.weight','fc2.bias'];"
            "print('ok')") % (os.path.join(ROOT, "sudoku-vision_amd", "cv"), os.path.join(ROOT, "sudoku-vision_amd", "ml"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_winograd_stream_schedule():
    """The producer schedule of k_conv_features_wstream (csrc/k3_cnn.hip): checked exhaustively for every run length a
    workgroup can get.  Invariants: a tile is transformed only after its cell's conv1; conv1(c) finds its input staged;
    a conv1 plane (double-buffered by cell parity) is never overwritten while tiles of the cell it held are still to be
    transformed in the same step; staging never writes the input buffer conv1 is reading."""
    def need(m, ncell):
        return min(ncell - 1, (16 * (m + 2) + 15) // 49)
    for ncell in list(range(1, 200)) + [1000]:
        nm = (ncell * 49 + 15) // 16
        conv_done = staged = min(1, ncell - 1)
        assert min(ncell - 1, 15 // 49) <= conv_done                       # M tile 0 (prologue transform)
        for m in range(nm):
            read_cells = set()
            if m + 1 < nm:
                lo, hi = (16 * (m + 1)) // 49, min(ncell - 1, (16 * (m + 1) + 15) // 49)
                assert hi <= conv_done
                read_cells = set(range(lo, hi + 1))
            cc = need(m, ncell)
            if cc > conv_done:
                assert cc == conv_done + 1 and cc <= staged
                assert (cc - 2) not in read_cells
                conv_done = cc
            sc = need(m + 1, ncell)
            if sc > staged:
                assert sc == staged + 1
                assert sc - 2 <= conv_done and not (cc == conv_done and (sc & 1) == (cc & 1) and sc != cc and cc > staged)
                assert (sc & 1) != (cc & 1) or sc == cc
                staged = sc
        assert conv_done == ncell - 1


def test_check_constraints_messages():
    """The harness's constraint messages (pipeline/run.py:205-241): format and the 'previous occurrence' rule."""
    from sudoku_vision_amd.pipeline import check_constraints
    g = [[0] * 9 for _ in range(9)]
    assert check_constraints(g) == []
    g[0][1] = g[0][4] = g[0][8] = 7                 # three 7s in row 1 (boxes 1, 2, 3: no box clash)
    g[3][2] = g[6][2] = 5                           # two 5s in column 3
    g[4][4] = g[5][5] = 9                           # two 9s in the centre box only
    assert check_constraints(g) == [
        "Row 1: duplicate 7 at columns 2 and 5", "Row 1: duplicate 7 at columns 5 and 9",
        "Column 3: duplicate 5 at rows 4 and 7",
        "Box (2,2): duplicate 9"]
    g2 = [[0] * 9 for _ in range(9)]
    g2[0][0] = g2[1][1] = g2[2][2] = 4              # box (1,1) three times
    assert check_constraints(g2) == ["Box (1,1): duplicate 4", "Box (1,1): duplicate 4"]
