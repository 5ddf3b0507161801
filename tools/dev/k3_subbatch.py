#!/usr/bin/env python3
"""Does the 260-MB feature round trip between k_conv_features and k_fc_head cost anything?  K2+K3 on 256 frames as one batch against 4 x 64 and
8 x 32 frames (the feature slab of a sub-batch, 65 / 33 MB, then fits the 256-MB Infinity Cache), per-kernel HIP-event times."""
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
out = {"logits": torch.empty((256, 81, 10), dtype=torch.float32, device="cuda"), "digits": torch.empty((256, 81), dtype=torch.uint8, device="cuda"),
       "conf": torch.empty((256, 81), dtype=torch.float32, device="cuda")}
ctx.reserve(256 * 81)
for sub in (256, 64, 32, 256, 64, 32):
    def step():
        for s in range(0, 256, sub):
            ctx.frames_to_digits(frames[s:s + sub], minv[s:s + sub], out={k: v[s:s + sub] for k, v in out.items()})
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    ctx.timing_begin()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        step()
    b.record()
    torch.cuda.synchronize()
    t = ctx.timing_end()
    print(f"sub-batch {sub:3d}: {a.elapsed_time(b) / 50:.4f} ms per 256 frames; per-kernel sums per 256 frames:",
          {k: round(v[0] / 50, 4) for k, v in t.items() if v[1]})
