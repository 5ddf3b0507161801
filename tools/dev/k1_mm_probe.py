#!/usr/bin/env python3
"""Matrix-pipe K1: time per 256 frames for both algorithms, ambiguous-pixel count, equality of the two kernels' outputs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for a in ("1", "0"):
        subprocess.run([sys.executable, __file__, a], env=dict(os.environ, SV_K1_ALGO=a))
    sys.exit(0)
import torch  # noqa: E402
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

ctx = sva.default_context()
frames = synth_frames(256, 1080, 1920, seed=1234, device="cuda")[0]
out = torch.empty((256, 1080, 1920), dtype=torch.uint8, device="cuda")
for _ in range(30):
    ctx.preprocess(frames, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    ctx.preprocess(frames, out=out)
b.record()
torch.cuda.synchronize()
amb, cap = ctx.preprocess_stats()
import hashlib  # noqa: E402
print(f"SV_K1_ALGO={sys.argv[1]}: {a.elapsed_time(b) / 30:.4f} ms per 256 frames; ambiguous {amb} of {256 * 1080 * 1920} px ({amb / (256 * 1080 * 1920):.2e}), cap {cap}; "
      f"sha {hashlib.sha256(out[:8].cpu().numpy().tobytes()).hexdigest()[:16]}")
