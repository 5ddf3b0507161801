"""Timing helper, not a test: µs per 1080p frame of the host corner search (sv_find_grid_corners_bits_batch) on one thread, on despeckled
binaries of synthetic frames made on the CPU with the oracle.  Lives under tests/ because it uses the oracle to make its inputs.

    python tests/perf_host_search.py [n_frames] [repeats] [other libsudokuvision_hip.so to time instead (A/B against an older build)]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cache = f"/tmp/perf_host_search_{n}.npy"
    if os.path.exists(cache):
        bits = np.load(cache)
    else:
        import sv_oracle as o
        from test_gpu_parity import _despeckle_np
        from sudoku_vision_amd.synth import synth_frames
        frames, _, _ = synth_frames(n, 1080, 1920, seed=11)
        bins = [_despeckle_np(o.preprocess_for_grid_detection(f.numpy())) for f in frames]
        bits = np.stack([np.packbits(b, axis=1, bitorder="little").view(np.uint32) for b in bins])
        np.save(cache, bits)
    if len(sys.argv) > 3:
        from sudoku_vision_amd import _native
        _native.LIB_PATH = sys.argv[3]
    from sudoku_vision_amd import host
    from test_host_contours import _pack_sparse_np
    cap = 1080 * 60 // 3                                               # the pipeline's record capacity
    recs = np.stack([_pack_sparse_np(host, b, cap) for b in bits])
    print("non-zero words per frame:", [int((b != 0).sum()) for b in bits])
    for name, data, fn in (("dense ", np.ascontiguousarray(np.tile(bits, (reps, 1, 1))), host.find_grid_corners_bits_batch),
                           ("sparse", np.ascontiguousarray(np.tile(recs, (reps, 1))), host.find_grid_corners_sparse_batch)):
        fn(data[:n], 1080, 1920, threads=1)
        best = 1e9
        for _ in range(15):
            t0 = time.perf_counter()
            c, f = fn(data, 1080, 1920, threads=1)
            best = min(best, time.perf_counter() - t0)
        print(f"{name} {best / len(data) * 1e6:.1f} us/frame   found {int((f == 1).sum())}/{len(data)}   checksum {int(c.astype(np.int64).sum())}")


if __name__ == "__main__":
    main()
