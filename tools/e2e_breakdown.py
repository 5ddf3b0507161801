#!/usr/bin/env python3
"""Where the end-to-end (host corner search in the loop) time goes.  Each stage of FramePipeline alone on 64-frame chunks of the bench
workload, then the pipeline itself, for the dense and the sparse hand-over:  python tools/e2e_breakdown.py [host_threads [chunk [depth]]]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd import host  # noqa: E402
from sudoku_vision_amd.pipeline import FramePipeline, host_cpu_budget  # noqa: E402
from sudoku_vision_amd.synth import random_state_dict, synth_frames  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, host_cpu_budget() - 2)
ctx = sva.default_context()
ctx.load_state_dict(random_state_dict(1234))
frames, corners, _ = synth_frames(256, 1080, 1920, seed=1234, device="cuda")
H, W, CH = 1080, 1920, int(sys.argv[2]) if len(sys.argv) > 2 else 64
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def gpu_ms(fn, reps=30):
    for _ in range(30):                     # the chip needs tens of milliseconds of load to reach its sustained clock
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for sparse in (False, True):
    pipe = FramePipeline(ctx, H, W, chunk=CH, host_threads=threads, sparse=sparse, depth=DEPTH)
    f = frames[:CH]
    slot = 0
    t_k1 = gpu_ms(lambda: ctx.preprocess_bits(f, out=pipe.dev_bits[slot]))
    raw_bits = ctx.preprocess_bits(f)
    t_desp = gpu_ms(lambda: ctx.despeckle_bits(pipe.dev_bits[slot].copy_(raw_bits))) - gpu_ms(lambda: pipe.dev_bits[slot].copy_(raw_bits))
    ctx.despeckle_bits(pipe.dev_bits[slot].copy_(raw_bits))
    payload = pipe.dev_bits[slot]
    t_pack = 0.0
    if sparse:
        t_pack = gpu_ms(lambda: ctx.pack_sparse_bits(pipe.dev_bits[slot], pipe.dev_rec[slot]))
        payload = pipe.dev_rec[slot]
    t_d2h = gpu_ms(lambda: pipe.pinned[slot].copy_(payload, non_blocking=True))
    torch.cuda.synchronize()
    arr = pipe.pinned[slot].numpy()
    t0 = time.perf_counter()
    for _ in range(10):
        if sparse:
            c, found = host.find_grid_corners_sparse_batch(arr, H, W, 0.1, 0.02, threads)
        else:
            c, found = host.find_grid_corners_bits_batch(arr, H, W, 0.1, 0.02, threads)
    t_search = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        minv, ok = sva.Context.corners_to_minv_batch(c.astype(np.float32))
    t_minv = (time.perf_counter() - t0) / 10 * 1e3
    md = ctx.minv_to_device(minv.reshape(CH, 9))
    out = {"logits": torch.empty((CH, 81, 10), dtype=torch.float32, device="cuda"), "digits": torch.empty((CH, 81), dtype=torch.uint8, device="cuda"),
           "conf": torch.empty((CH, 81), dtype=torch.float32, device="cuda")}
    t_cls = gpu_ms(lambda: ctx.frames_to_digits(f, md, out=out))
    pipe.run(frames)
    torch.cuda.synchronize()
    trials, throttled = [], []

    def cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
        except OSError:
            return {}

    for _ in range(6):                      # the box gives the job a CPU quota (cgroup cpu.max): let it refill, and report every trial
        time.sleep(0.5)
        st0 = cpu_stat()
        t0 = time.perf_counter()
        pipe.run(frames, repeat=8)
        torch.cuda.synchronize()
        trials.append(time.perf_counter() - t0)
        st1 = cpu_stat()
        throttled.append((st1.get("nr_throttled", 0) - st0.get("nr_throttled", 0), (st1.get("throttled_usec", 0) - st0.get("throttled_usec", 0)) // 1000,
                          (st1.get("usage_usec", 0) - st0.get("usage_usec", 0)) // 1000))
    dt = min(trials)
    gpu = t_k1 + t_desp + t_pack + t_cls
    print(f"{'sparse' if sparse else 'dense '} hand-over, {threads} host threads, depth {DEPTH}, per {CH}-frame chunk [ms]: K1 (bits) {t_k1:.3f}  despeckle (bits, in place) {t_desp:.3f}  pack {t_pack:.3f}  "
          f"D2H {t_d2h:.3f} ({payload.numel() * payload.element_size() / CH / 1e3:.0f} KB/frame)  search {t_search:.3f}  minv {t_minv:.3f}  K2+K3 {t_cls:.3f}  "
          f"| GPU sum {gpu:.3f} -> {CH / gpu * 1e3:.0f} f/s, host {t_search + t_minv:.3f} -> {CH / (t_search + t_minv) * 1e3:.0f} f/s, D2H -> {CH / t_d2h * 1e3:.0f} f/s "
          f"| pipeline {256 * 8 / dt:.0f} f/s = {dt / (256 * 8 / CH) * 1e3:.3f} ms per chunk; found {int(found.sum())}/{CH}, fallbacks {pipe.dense_fallbacks}; trials {[round(256 * 8 / t) for t in trials]}; (throttle events, throttled ms, cpu ms) per trial {throttled}")
