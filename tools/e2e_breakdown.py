#!/usr/bin/env python3
"""Where the end-to-end (host corner search in the loop) time goes: wall time of the C search call per chunk, of the
minv computation, and of the whole run.  python tools/e2e_breakdown.py [host_threads]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd import host  # noqa: E402
from sudoku_vision_amd.pipeline import FramePipeline  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402
import cnn_oracle  # noqa: E402

from sudoku_vision_amd.pipeline import host_cpu_budget  # noqa: E402
threads = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, host_cpu_budget() - 2)
ctx = sva.default_context()
ctx.load_state_dict(cnn_oracle.random_state_dict(1234))
frames, corners, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
pipe = FramePipeline(ctx, 1080, 1920, chunk=64, host_threads=threads)
pipe.run(frames)
torch.cuda.synchronize()
t_search, t_minv, t_wait = [], [], []
orig = host.find_grid_corners_bits_batch
orig_minv = sva.Context.corners_to_minv


def timed(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    t_search.append(time.perf_counter() - t)
    return r


def timed_minv(*a, **k):
    t = time.perf_counter()
    r = orig_minv(*a, **k)
    t_minv.append(time.perf_counter() - t)
    return r


host.find_grid_corners_bits_batch = timed
sva.Context.corners_to_minv = staticmethod(timed_minv)
t0 = time.perf_counter()
pipe.run(frames, repeat=8)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = 256 * 8
print(f"threads {threads}: {n / dt:.0f} frames/s; per 64-frame chunk: total {dt / 32 * 1e3:.2f} ms, C search {np.mean(t_search) * 1e3:.2f} ms "
      f"(min {np.min(t_search) * 1e3:.2f}, max {np.max(t_search) * 1e3:.2f}), minv {np.mean(t_minv) * 1e3:.3f} ms; os.cpu_count {os.cpu_count()}, "
      f"affinity {len(os.sched_getaffinity(0))}")
