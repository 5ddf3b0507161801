#!/usr/bin/env python3
"""Why is bench.py's first end-to-end region slower than the next ones?  Replays the bench's sequence and prints per-call times and, for the
first timed call, the completion time of every chunk."""
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
n = 256
frames, corners, _ = synth_frames(n, 1080, 1920, seed=1234, device="cuda")
minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners))
out = {"logits": torch.empty((n, 81, 10), dtype=torch.float32, device="cuda"), "digits": torch.empty((n, 81), dtype=torch.uint8, device="cuda"),
       "conf": torch.empty((n, 81), dtype=torch.float32, device="cuda")}
binary = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
for _ in range(220):                                  # the device-only phase of the bench
    ctx.preprocess(frames, out=binary)
    ctx.frames_to_digits(frames, minv, out=out)
torch.cuda.synchronize()
threads = max(1, min(16, host_cpu_budget()) - 2)
pipe = FramePipeline(ctx, 1080, 1920, chunk=256, host_threads=threads)
t_w = time.perf_counter()
pipe.run(frames, out=out, total=n * 20)
while time.perf_counter() - t_w < 0.3:
    pipe.run(frames, out=out, total=8 * n)
torch.cuda.synchronize()
stamps = []
orig = pipe._search


def search(slot, m, ev):
    r = orig(slot, m, ev)
    stamps.append(time.perf_counter())
    return r


pipe._search = search
for rep in range(5):
    stamps.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(frames, out=out, total=n * 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gaps = [round((b - a) * 1e3, 2) for a, b in zip([t0] + stamps[:-1], stamps)]
    print(f"call {rep}: {n * 100 / dt:.0f} frames/s; search-done gaps (ms) first 12: {gaps[:12]} ... max {max(gaps):.2f} at chunk {gaps.index(max(gaps))}, "
          f"gaps > 3 ms: {[(i, g) for i, g in enumerate(gaps) if g > 3.0][:8]}", flush=True)
