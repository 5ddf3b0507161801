#!/usr/bin/env python3
"""Find where a slow end-to-end trial loses its time: timestamps of every chunk's stages (host side) for the slowest of N trials."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd import host, pipeline  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict, synth_frames  # noqa: E402

ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1234))
frames = synth_frames(256, 1080, 1920, seed=1234, device="cuda")[0]
pipe = pipeline.FramePipeline(ctx, 1080, 1920, chunk=64, host_threads=14)
log = []
orig_search = pipe._search


def traced_search(slot, m, ev):
    t0 = time.perf_counter()
    ev.synchronize()
    t1 = time.perf_counter()
    r = orig_search(slot, m, ev)
    t2 = time.perf_counter()
    log.append(("search", slot, t0, t1, t2))
    return r


pipe._search = traced_search
orig_f2d = ctx.frames_to_digits
orig_pre = ctx.preprocess


def traced_f2d(*a, **k):
    t0 = time.perf_counter()
    r = orig_f2d(*a, **k)
    log.append(("f2d", 0, t0, time.perf_counter(), 0))
    return r


def traced_pre(*a, **k):
    t0 = time.perf_counter()
    r = orig_pre(*a, **k)
    log.append(("pre", 0, t0, time.perf_counter(), 0))
    return r


ctx.frames_to_digits = traced_f2d
ctx.preprocess = traced_pre
pipe.run(frames)
torch.cuda.synchronize()
worst = None
for trial in range(12):
    time.sleep(0.3)
    log.clear()
    t0 = time.perf_counter()
    pipe.run(frames, repeat=8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"trial {trial}: {2048 / dt:.0f} f/s")
    if worst is None or dt > worst[0]:
        worst = (dt, t0, list(log))
dt, t0, lg = worst
print(f"slowest trial: {dt * 1e3:.1f} ms")
for kind, slot, a, b, c in sorted(lg, key=lambda e: e[2]):
    if kind == "search":
        print(f"  {1e3 * (a - t0):8.2f} search slot {slot}: waited for the copy {1e3 * (b - a):7.2f} ms, searched {1e3 * (c - b):6.2f} ms")
    else:
        print(f"  {1e3 * (a - t0):8.2f} {kind} call took {1e3 * (b - a):6.2f} ms")
