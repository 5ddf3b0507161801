#!/usr/bin/env python3
"""Time of sv_despeckle_bits (two passes, in place) and sv_pack_sparse_bits on the K1 bit images of 256 synthetic 1080p frames; the restoring copy of the
66-MB bit image is timed alone and subtracted.  Optional argument: another libsudokuvision_hip.so (A/B against a variant build)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from sudoku_vision_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd import host  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

ctx = sva.default_context()
frames, _, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
src = ctx.preprocess_bits(frames)
work = torch.empty_like(src)
records = torch.empty((256, host.sparse_bits_record_bytes(1080, 1920, 1080 * 60 // 3)), dtype=torch.uint8, device="cuda")


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_copy = t(lambda: work.copy_(src))
t_both = t(lambda: ctx.despeckle_bits(work.copy_(src)))
ctx.despeckle_bits(work.copy_(src))
t_pack = t(lambda: ctx.pack_sparse_bits(work, records))
print(f"copy {t_copy:.4f} ms   despeckle (2 passes) {t_both - t_copy:.4f} ms   pack {t_pack:.4f} ms   (per 256 frames)   checksum {int(work.sum())}")
