#!/usr/bin/env python3
"""Same-box comparison of FramePipeline settings (sparse record capacity, chunk, depth, host threads): steady-state frames/s over 100 steps."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.pipeline import FramePipeline, host_cpu_budget  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict, synth_frames  # noqa: E402

ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1234))
frames, corners, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
threads = max(1, min(16, host_cpu_budget()) - 2)
configs = [dict(chunk=128, sparse=True), dict(chunk=128, sparse=21600), dict(chunk=256, sparse=True), dict(chunk=256, sparse=21600), dict(chunk=128, sparse=True, depth=4),
           dict(chunk=128, sparse=True, depth=7), dict(chunk=128, sparse=True, host_threads=10), dict(chunk=128, sparse=True, host_threads=12), dict(chunk=64, sparse=True, depth=8)]
res = {i: [] for i in range(len(configs))}
pipes = [FramePipeline(ctx, 1080, 1920, **{"host_threads": threads, **c}) for c in configs]
for p in pipes:
    p.run(frames, total=256 * 20)
for rep in range(3):
    for i, p in enumerate(pipes):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = p.run(frames, total=256 * 100)
        torch.cuda.synchronize()
        res[i].append(256 * 100 / (time.perf_counter() - t))
for i, c in enumerate(configs):
    print(c, [round(v) for v in res[i]], "fallbacks", pipes[i].dense_fallbacks, flush=True)
