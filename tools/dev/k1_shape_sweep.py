#!/usr/bin/env python3
"""K1 launch shape: bands per frame (waves per launch) for 256- and 128-frame launches, bit image output."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

lib = sva._native.lib()
ctx = sva.default_context()
frames, _, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
ref = ctx.preprocess_bits(frames).clone()
ref_b = ctx.preprocess(frames).clone()


def ms(fn, reps=40):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


bits = torch.empty_like(ref)
byts = torch.empty_like(ref_b)
for waves in (4096, 6144, 8192, 10240, 12288, 16384, 20480):
    pass  # (the sweep used a development knob for the wave count; the result is the constant in march_shape)
    assert torch.equal(ctx.preprocess_bits(frames), ref) and torch.equal(ctx.preprocess(frames), ref_b)
    print(f"waves {waves:6d}: bits/256 {ms(lambda: ctx.preprocess_bits(frames, out=bits)):.4f}  bytes/256 {ms(lambda: ctx.preprocess(frames, out=byts)):.4f}  "
          f"bits/128 {ms(lambda: ctx.preprocess_bits(frames[:128], out=bits[:128])):.4f}  bits/64 {ms(lambda: ctx.preprocess_bits(frames[:64], out=bits[:64])):.4f} ms", flush=True)
