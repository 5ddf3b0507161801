#!/usr/bin/env python3
"""K3 (conv + fc, HIP-event kernel times) over batch sizes, for choosing where k_fc_head_h2p takes over from k_fc_head_h2.  Argument: the library to time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from sudoku_vision_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict  # noqa: E402

ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1))
x = torch.randint(0, 256, (32768, 28, 28), dtype=torch.uint8, device="cuda")
ctx.reserve(32768)
out = []
for B in (81, 324, 1296, 2592, 5184, 8192, 10368, 13000, 16384, 20736, 32768):
    for _ in range(20):
        ctx.cnn_forward(x[:B])
    torch.cuda.synchronize()
    ctx.timing_begin()
    for _ in range(50):
        ctx.cnn_forward(x[:B])
    torch.cuda.synchronize()
    k = ctx.timing_end()
    out.append(f"B={B}: fc {k['k_fc_head'][0] / k['k_fc_head'][1]:.4f} conv {k['k_conv_features'][0] / k['k_conv_features'][1]:.4f}")
print(" | ".join(out))
