#!/usr/bin/env python3
"""Per-phase s_memtime stamps of the conv consumers.  Needs a development build of the library (k3_cnn_h2.hip compiled with -DSV_DEV):
    cd sudoku-vision_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DSV_DEV -c k3_cnn_h2.hip -o /tmp/k3_dev.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dev/libsv_dev.so sv_api.o host_contours.o host_solver.o host_jpeg.o k1_threshold.o k2_warp_cells.o \
          k3_cnn.o k3_cnn_bf16.o /tmp/k3_dev.o k4_despeckle.o k5_jpeg.o
Round-3 result (cycles per cell, consumer waves): tiles 8042 (7 M tiles) / 7533 (6), barrier wait 91 / 608; without conv1 7103 / 6270; the
producers alone 4275.  A variant with one barrier per PAIR of cells and alternating tile parity (13 tiles per wave per pair, four cells'
planes in LDS) was bit-identical and 1 % faster (0.3515 vs 0.3525 ms): the loss is inside the tile body, not at the barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict  # noqa: E402

sva._native.LIB_PATH = os.path.join(ROOT, "tools", "dev", "libsv_dev.so")
lib = sva._native.lib()
ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1234))
cells = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (20736, 28, 28)).astype(np.uint8)).cuda()
for pair in (0,):
    for _ in range(20):
        ctx.cnn_forward(cells)
    torch.cuda.synchronize()
    for bits in (16, 16 | 1, 16 | 2):
        print(f"--- pair {pair} ablate {bits} (1 = no conv1 tiles, 2 = no conv2 tiles)", flush=True)
        lib.sv_dev_set_h2_ablate(ctx._h, bits)
        ctx.cnn_forward(cells)
        torch.cuda.synchronize()
    lib.sv_dev_set_h2_ablate(ctx._h, 0)
