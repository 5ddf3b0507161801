#!/usr/bin/env python3
"""Cross-step overlap: K2 of step i+1 on a second stream, released when K1 of step i has finished, so that it runs in the shadow of step i's
conv (no registers free: it waits) and fc (memory-bound, half the register file).  Steady-state time per 256-frame step against the serial order."""
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
cells = [torch.empty((256, 81, 28, 28), dtype=torch.uint8, device="cuda") for _ in range(2)]
ctx.reserve(256 * 81)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
lib, C = sva._native.lib(), sva._native.C


def k2(into):
    sva._native.check(lib.sv_warp_cells_u8(ctx._h, C.c_void_p(frames.data_ptr()), 256, 1080, 1920, 3 * 1920, 3 * 1920 * 1080, C.c_void_p(minv.data_ptr()),
                                           C.c_void_p(into.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "k2")


def serial(steps):
    with torch.cuda.stream(sA):
        for _ in range(steps):
            ctx.preprocess(frames, out=binary)
            k2(cells[0])
            ctx.cnn_forward(cells[0].view(-1, 28, 28), want_digits=True)


def overlapped(steps):
    ev_k2 = [None, None]
    ev_k3 = [None, None]
    with torch.cuda.stream(sB):
        k2(cells[0])
        ev_k2[0] = torch.cuda.Event(); ev_k2[0].record(sB)
    for i in range(steps):
        with torch.cuda.stream(sA):
            ctx.preprocess(frames, out=binary)
            ev_k1 = torch.cuda.Event(); ev_k1.record(sA)
        if i + 1 < steps:
            with torch.cuda.stream(sB):
                sB.wait_event(ev_k1)                                   # not beside K1 (that pairing is slower)
                if ev_k3[(i + 1) & 1] is not None:
                    sB.wait_event(ev_k3[(i + 1) & 1])                  # the cell buffer is free again
                k2(cells[(i + 1) & 1])
                ev_k2[(i + 1) & 1] = torch.cuda.Event(); ev_k2[(i + 1) & 1].record(sB)
        with torch.cuda.stream(sA):
            sA.wait_event(ev_k2[i & 1])
            ctx.cnn_forward(cells[i & 1].view(-1, 28, 28), want_digits=True)
            ev_k3[i & 1] = torch.cuda.Event(); ev_k3[i & 1].record(sA)


for name, fn in (("serial", serial), ("K2(i+1) under K3(i)", overlapped)) * 2:
    fn(60)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(sA)
    fn(100)
    sA.wait_stream(sB)
    b.record(sA)
    torch.cuda.synchronize()
    print(f"{name:22s}: {a.elapsed_time(b) / 100:.4f} ms per 256-frame step")
