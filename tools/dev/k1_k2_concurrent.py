#!/usr/bin/env python3
"""K1 is VALU-issue bound, K2 a latency-bound gather, and in device-only mode they are independent (both read the frame): do they overlap when
launched on two streams?  Step time K1 -> K2 -> K3 in one stream against (K1 || K2) -> K3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict, synth_frames  # noqa: E402

ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1234))
frames, corners, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners).reshape(256, 9))
binary = torch.empty((256, 1080, 1920), dtype=torch.uint8, device="cuda")
ctx.reserve(256 * 81)
s2 = torch.cuda.Stream()
cur = torch.cuda.current_stream()


def serial():
    ctx.preprocess(frames, out=binary)
    cells = ctx.warp_cells(frames, minv)
    return ctx.cnn_forward(cells.view(-1, 28, 28), want_digits=True)


def overlapped():
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        cells = ctx.warp_cells(frames, minv)
        cells.record_stream(cur)
    ctx.preprocess(frames, out=binary)
    cur.wait_stream(s2)
    return ctx.cnn_forward(cells.view(-1, 28, 28), want_digits=True)


def k2_first():
    cells = ctx.warp_cells(frames, minv)
    ctx.preprocess(frames, out=binary)
    return ctx.cnn_forward(cells.view(-1, 28, 28), want_digits=True)


for name, fn in (("K1 -> K2 -> K3", serial), ("(K1 || K2) -> K3", overlapped), ("K2 -> K1 -> K3", k2_first)) * 2:
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:18s}: {a.elapsed_time(b) / 100:.4f} ms per 256 frames")
