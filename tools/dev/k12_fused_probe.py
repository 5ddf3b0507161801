#!/usr/bin/env python3
"""configs[4] fused threshold + warp launch against the two separate launches: outputs equal, time per 256 frames."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

ctx = sva.default_context()
frames, corners, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners).reshape(256, 9))
binary = torch.empty((256, 1080, 1920), dtype=torch.uint8, device="cuda")
b2, c2 = torch.empty_like(binary), torch.empty((256, 81, 28, 28), dtype=torch.uint8, device="cuda")


def separate():
    ctx.preprocess(frames, out=binary)
    return ctx.warp_cells(frames, minv)


def fused():
    ctx.preprocess_and_warp_cells(frames, minv, binary=b2, cells=c2)


cells = separate()
fused()
torch.cuda.synchronize()
print("binary equal", torch.equal(binary, b2), "cells equal", torch.equal(cells, c2))
if len(sys.argv) > 1:
    which = {"separate": separate, "fused": fused}[sys.argv[1]]
    for _ in range(3):
        which()
    torch.cuda.synchronize()
    sys.exit(0)
for name, fn in (("K1 then K2", separate), ("fused launch", fused)) * 2:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:14s}: {a.elapsed_time(b) / 50:.4f} ms per 256 frames")
