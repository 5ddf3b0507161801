#!/usr/bin/env python3
"""Host search alone, 64 despeckled synthetic 1080p frames, 14 threads: dense bit images against sparse records (expansion + mask-guided scan)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd import host, pipeline  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ctx = sva.default_context()
frames = synth_frames(64, 1080, 1920, seed=1234, device="cuda")[0]
bits = ctx.despeckle_bits(ctx.preprocess_bits(frames))
stride = host.sparse_bits_record_bytes(1080, 1920, 1080 * 60 // 2)
rec = ctx.pack_sparse_bits(bits, torch.empty((64, stride), dtype=torch.uint8, device="cuda"))
dense_h, rec_h = bits.cpu().numpy(), rec.cpu().numpy()
cpus = pipeline.gpu_local_cpus(torch.device("cuda", 0))
if cpus:
    host.set_pool_affinity(cpus)
    os.sched_setaffinity(0, cpus)
for name, fn in (("dense ", lambda: host.find_grid_corners_bits_batch(dense_h, 1080, 1920, 0.1, 0.02, threads)),
                 ("sparse", lambda: host.find_grid_corners_sparse_batch(rec_h, 1080, 1920, 0.1, 0.02, threads))) * 2:
    d = []
    for _ in range(1500):
        t = time.perf_counter()
        fn()
        d.append(time.perf_counter() - t)
    d = np.array(d) * 1e3
    print(f"{name}: median {np.median(d):.3f} ms per 64 frames, p90 {np.percentile(d, 90):.3f}, max {d.max():.2f}")
