#!/usr/bin/env python3
"""Generates tests/golden/solver_golden.npz by running the REFERENCE solver (solver/src/sudoku.c, compiled as is into
oracle/_ref/libsudoku_ref.so by `make -C oracle ref`) on seeded puzzles.  Run in the build container only."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def puzzles(seed=7, n=160):
    rs = np.random.RandomState(seed)
    base = np.array([[(3 * (r % 3) + r // 3 + c) % 9 + 1 for c in range(9)] for r in range(9)], np.int32)   # a valid filled grid
    out = []
    for k in range(n):
        g = base.copy()
        g = (rs.permutation(9) + 1)[g - 1]                                                   # relabel digits
        for b in range(3):                                                                   # shuffle rows/cols inside bands
            p = rs.permutation(3)
            g[3 * b:3 * b + 3] = g[3 * b + p]
            p = rs.permutation(3)
            g[:, 3 * b:3 * b + 3] = g[:, 3 * b + p]
        clues = rs.randint(0, 62)                                                            # 0..61 clues: unique, multiple and trivial cases
        mask = np.zeros(81, bool)
        mask[rs.permutation(81)[:clues]] = True
        g = np.where(mask.reshape(9, 9), g, 0)
        kind = k % 8
        if kind == 6:                                                                        # invalid: a duplicate / out of range value
            r, c = rs.randint(0, 9, 2)
            g[r, c] = g[r, (c + 1) % 9] if g[r, (c + 1) % 9] else 10
            if g[r, c] == 0:
                g[r, c] = 12
        if kind == 7:                                                                        # valid-looking but wrong digit (often unsolvable)
            idx = np.flatnonzero(g.ravel())
            if len(idx):
                i = idx[rs.randint(len(idx))]
                g.ravel()[i] = (g.ravel()[i] % 9) + 1
        out.append(g)
    return np.stack(out).astype(np.int32)


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsudoku_ref.so"))
    P = puzzles()
    codes, sols = [], []
    for g in P:
        w = np.ascontiguousarray(g.copy())
        codes.append(lib.solve_sudoku(w.ctypes.data_as(C.c_void_p)))
        sols.append(w)
    codes = np.array(codes, np.int32)
    print("codes:", {int(c): int((codes == c).sum()) for c in np.unique(codes)})
    np.savez_compressed(os.path.join(HERE, "solver_golden.npz"), puzzles=P, codes=codes, solutions=np.stack(sols).astype(np.int32))


if __name__ == "__main__":
    main()
