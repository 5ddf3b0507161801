"""CPU: scope row N4, the in-process solver (csrc/host_solver.cpp) against goldens produced by the reference's own
solver source (tests/golden/make_solver_goldens.py) and, where oracle/_ref exists, against that library live."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    import sudoku_vision_amd as sva
    sva._native.lib()
    return sva.host


def _check_solution(p, s):
    assert ((p == 0) | (p == s)).all()
    for u in list(s) + list(s.T) + [s[r:r + 3, c:c + 3].ravel() for r in (0, 3, 6) for c in (0, 3, 6)]:
        assert sorted(u) == list(range(1, 10))


def test_against_reference_goldens(host, golden_dir):
    g = np.load(os.path.join(golden_dir, "solver_golden.npz"))
    seen = set()
    for p, code, sol in zip(g["puzzles"], g["codes"], g["solutions"]):
        c, s = host.solve_sudoku(p)
        assert c == code
        seen.add(int(code))
        if code == 1:
            assert (s == sol).all()          # the same grid as the reference, also when several solutions exist
            _check_solution(p, s)
        elif (p >= 0).all() and (p <= 255).all():
            assert (s == p).all()
    assert seen == {-1, 0, 1}


def test_live_against_reference_library(host):
    ref = os.path.join(ROOT, "oracle", "_ref", "libsudoku_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (no reference tree on this box)")
    lib = C.CDLL(ref)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_solver_goldens.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    for p in mk.puzzles(seed=99, n=120):
        w = np.ascontiguousarray(p.copy())
        code = lib.solve_sudoku(w.ctypes.data_as(C.c_void_p))
        c, s = host.solve_sudoku(p)
        assert c == code and (code != 1 or (s == w).all())


def test_run_solver_mirror(host):
    from sudoku_vision_amd.pipeline import run_solver
    puzzle = [[5, 3, 0, 0, 7, 0, 0, 0, 0], [6, 0, 0, 1, 9, 5, 0, 0, 0], [0, 9, 8, 0, 0, 0, 0, 6, 0], [8, 0, 0, 0, 6, 0, 0, 0, 3],
              [4, 0, 0, 8, 0, 3, 0, 0, 1], [7, 0, 0, 0, 2, 0, 0, 0, 6], [0, 6, 0, 0, 0, 0, 2, 8, 0], [0, 0, 0, 4, 1, 9, 0, 0, 5],
              [0, 0, 0, 0, 8, 0, 0, 7, 9]]
    ok, sol = run_solver(puzzle)
    assert ok and sol[0] == [5, 3, 4, 6, 7, 8, 9, 1, 2] and sol[8] == [3, 4, 5, 2, 8, 6, 1, 7, 9]
    bad = [row[:] for row in puzzle]
    bad[0][2] = 5
    assert run_solver(bad) == (False, bad)            # duplicates: the reference returns (False, grid)
    assert host.solve_sudoku(np.full((9, 9), 300))[0] == -1
