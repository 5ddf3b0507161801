#!/usr/bin/env python3
"""Prints the per-step timeline recorded by a build instrumented with tools/dev/instrument_conv.py."""
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "oracle"))
import cnn_oracle  # noqa: E402
import sudoku_vision_amd as sva  # noqa: E402

ctx = sva.default_context()
ctx.load_state_dict(cnn_oracle.random_state_dict(1234))
x = torch.randint(0, 256, (20736, 28, 28), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.cnn_forward(x)
torch.cuda.synchronize()
buf = np.zeros(64 * 8 * 6, np.uint64)
sva._native.lib().sv_debug_conv_trace(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(64, 8, 6).astype(np.int64)
for m in range(0, 12):
    c, p = t[m, 0], t[m, 4]
    print(m + 40, "cons: gemm %5d out %5d help %5d wait %5d | prod: xform %5d conv1 %5d wait %5d | step %d"
          % (c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], p[1] - p[0], p[2] - p[1], p[4] - p[3], t[m + 1, 0, 0] - c[0]))
d = t[1:, 0, 0] - t[:-1, 0, 0]
cons, prod = t[:, 0], t[:, 4]
print("mean step %.0f ticks" % d.mean())
print("consumer mean: gemm %.0f out %.0f help %.0f wait %.0f" % tuple((cons[:, i + 1] - cons[:, i]).mean() for i in range(4)))
print("producer mean: xform %.0f conv1 %.0f wait %.0f" % ((prod[:, 1] - prod[:, 0]).mean(), (prod[:, 2] - prod[:, 1]).mean(), (prod[:, 4] - prod[:, 3]).mean()))
