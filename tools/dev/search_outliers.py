#!/usr/bin/env python3
"""Duration distribution of the batch corner search alone (no GPU work in flight): are the 40-60 ms outliers of the end-to-end run the host's own?"""
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
bits = torch.empty((64, 1080, 60), dtype=torch.int32, device="cuda")
b = ctx.preprocess(frames)
ctx.despeckle(b, out=b, packed=bits)
pinned = torch.empty((64, 1080, 60), dtype=torch.int32).pin_memory()
pinned.copy_(bits)
torch.cuda.synchronize()
cpus = pipeline.gpu_local_cpus(torch.device("cuda", 0))
for label, arr in (("pinned", pinned.numpy()), ("pageable copy", pinned.numpy().copy())):
    for aff in (None, cpus):
        if aff:
            host.set_pool_affinity(aff)
            os.sched_setaffinity(0, aff)
        d = []
        for _ in range(3000):
            t = time.perf_counter()
            host.find_grid_corners_bits_batch(arr, 1080, 1920, 0.1, 0.02, threads)
            d.append(time.perf_counter() - t)
        d = np.array(d) * 1e3
        print(f"{label:14s} affinity {'node' if aff else 'none'}: median {np.median(d):.3f} ms  p99 {np.percentile(d, 99):.3f}  p99.9 {np.percentile(d, 99.9):.3f}  max {d.max():.2f}  "
              f"calls > 2 ms: {(d > 2).sum()}  > 20 ms: {(d > 20).sum()}")
