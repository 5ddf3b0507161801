#!/usr/bin/env python3
"""Benchmark of the frame -> digits hot path on MI355X (BASELINE.json metric: frames/s, 1080p -> 81 digits).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic 1080p frames already resident in HBM:
K1 (frame -> binary image for the host corner search) + K2 (frame + homography -> 81 cells) +
K3 (cells -> logits, digits).  Workload at every N: BASELINE.json configs[1] per GPU -- 256 synthetic
1080p frames, fp32 CNN -- with the generator's ground-truth corners (device-only figure; the host
corner search is reported separately once it is in the loop).  Frames shard by rank with no
collective (weak scaling); the only torch.distributed use is the timing barrier and a MAX of elapsed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per unit (restated in DESIGN.md)
BYTES_PER_FRAME = 9_916_344          # read 6,220,800 + <=1,555,200 ; write 2,073,600 + 63,504 + 3,240
K1_BYTES_PER_FRAME = 6_220_800 + 2_073_600
K2_BYTES_PER_FRAME = 1_555_200 + 63_504
CONV_FLOP_PER_CELL = 451_584 + 7_225_344
# what k_conv_features_wstream executes per cell: 1568 v_mfma_f32_16x16x4_f32 (Winograd F(2x2,3x3): 2.25x fewer multiplies than
# the direct 3x3 convolution the algorithmic figure counts) + conv1 on the VALU + the two Winograd transforms' adds
CONV_EXECUTED_FLOP_PER_CELL = 1568 * 2048 + 451_584 + 49 * 32 * 32 + 49 * 64 * 24
FC_FLOP_PER_CELL = 802_816 + 2_560
HBM_PEAK = 8.0e12                    # B/s, MI355X_MICROARCH.md
FP32_MFMA_PEAK = 157.3e12            # FLOP/s, v_mfma_f32_* (= fp32 vector peak)
BF16_MFMA_PEAK = 2.5e15              # FLOP/s dense, v_mfma_f32_*_bf16


def cpu_baseline(frames_host, corners, sd, threads):
    """The oracle (CPU port of the reference arithmetic) on a bounded sample: K1+K2 in C, one frame per
    thread; CNN with torch-CPU on all cells.  Baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import torch
    import cnn_oracle
    import sv_oracle
    sv_oracle.lib()
    torch.set_num_threads(threads)
    n = frames_host.shape[0]

    def one(i):
        sv_oracle.preprocess_for_grid_detection(frames_host[i])
        return sv_oracle.warp_cells(frames_host[i], corners[i])

    t0 = time.perf_counter()
    passes = 0
    with ThreadPoolExecutor(threads) as ex:
        while True:                                   # bounded sample: whole passes until >= 10 s of wall time
            cells = list(ex.map(one, range(n)))
            x = sv_oracle.cells_to_input(np.stack(cells).reshape(-1, 28, 28))[:, None]
            cnn_oracle.predict(sd, x)
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or passes >= 50:
                break
    return {"value": n * passes / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{passes} pass(es) over {n} of the benchmark's synthetic 1080p frames ({n * passes} frame evaluations, {dt:.1f} s): "
                      "C oracle K1+K2, one frame per thread, + torch-CPU DigitCNN on the same threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (headline); bf16 = configs[4]: conv2/fc1 on bf16 MFMA, digit-index parity only")
    ap.add_argument("--e2e-passes", type=int, default=8, help="passes over the pool with the host corner search in the loop (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import sudoku_vision_amd as sva
    from sudoku_vision_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SV_BENCH_REHEARSE=1: rehearse the multi-rank code path on a one-GPU box (all ranks on cuda:0, gloo instead of RCCL)
    rehearse = os.environ.get("SV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharding.init("gloo" if rehearse else "nccl")          # RCCL; only for the timing barrier and the MAX of elapsed
    dev = None if rehearse else torch.device("cuda", local_rank)

    from sudoku_vision_amd.synth import synth_frames
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnn_oracle

    ctx = sva.default_context()
    n = args.frames
    H, W = 1080, 1920
    frames, corners, _ = synth_frames(n, H, W, seed=1234 + rank, device="cuda")
    sd = cnn_oracle.random_state_dict(1234)        # random-init weights of the DigitCNN architecture
    ctx.load_state_dict(sd)
    ctx.reserve(n * 81)
    if args.precision == "bf16":
        ctx.set_precision(ctx.PREC_BF16)
    minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners))
    out = {"logits": torch.empty((n, 81, 10), dtype=torch.float32, device="cuda"),
           "digits": torch.empty((n, 81), dtype=torch.uint8, device="cuda"),
           "conf": torch.empty((n, 81), dtype=torch.float32, device="cuda")}

    def step():
        binary = ctx.preprocess(frames)              # K1: what the host corner search consumes
        ctx.frames_to_digits(frames, minv, out=out)  # K2 -> K3
        return binary

    def barrier():
        torch.cuda.synchronize()
        sharding.barrier(dev)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_kernel = ctx.timing_end()
    elapsed = sharding.max_over_ranks(elapsed, dev)

    # second figure: the same pool with the host corner search in the loop (K1 -> D2H -> CPU contours -> K2 -> K3)
    e2e = None
    if args.e2e_passes > 0:
        from sudoku_vision_amd.pipeline import FramePipeline, host_cpu_budget
        # the box's CPU share is 16 cores per GPU, enforced as a cgroup quota: stay two under it (this thread + the HIP runtime's)
        budget = host_cpu_budget()
        if budget > 16 * world:                      # no quota (or one far above the per-GPU share): split the host evenly over the ranks
            budget //= world
        host_threads = max(1, min(16, budget) - 2)
        pipe = FramePipeline(ctx, H, W, chunk=64 if n % 64 == 0 else 32, host_threads=host_threads)
        pipe.run(frames, out=out)                     # warm-up (page-locks, thread start)
        barrier()
        t1 = time.perf_counter()
        res_e2e = pipe.run(frames, out=out, repeat=args.e2e_passes)     # the pool streamed e2e_passes times, pipeline kept full
        barrier()
        dt = time.perf_counter() - t1
        dt = sharding.max_over_ranks(dt, dev)
        err = np.abs(res_e2e["corners"].astype(np.float32)[:, :, None, :] - corners[:, None, :, :]).sum(-1).min(-1).max()
        e2e = {"value": n * args.e2e_passes * world / dt, "unit": "frames/s", "host_threads_per_gpu": host_threads,
               "grids_found": int(res_e2e["found"].sum()), "of": n, "max_corner_error_px": float(err),
               "note": "K1 -> despeckle (exact speck filter) -> pinned D2H of the bit-packed binary (259 KB/frame over PCIe) -> C++ contour corner search on host threads -> K2 -> K3, 64-frame chunks triple-buffered"}

    if rank == 0:
        total_frames = n * args.steps * world
        fps = total_frames / elapsed
        cells = n * 81
        work = {"k_preprocess": ("hbm", K1_BYTES_PER_FRAME * n), "k_warp_cells": ("hbm", K2_BYTES_PER_FRAME * n),
                "k_conv_features": ("mfma", CONV_FLOP_PER_CELL * cells), "k_fc_head": ("mfma", FC_FLOP_PER_CELL * cells)}
        kernels = {}
        for name, (ms, cnt) in per_kernel.items():
            if not cnt:
                continue
            bound, units = work[name]
            avg = ms / cnt * 1e-3
            peak = HBM_PEAK if bound == "hbm" else (BF16_MFMA_PEAK if args.precision == "bf16" else FP32_MFMA_PEAK)
            ach = units / avg
            kernels[name] = {"bound": bound, "avg_ms": ms / cnt, "launches": cnt,
                             "achieved": ach / (1e9 if bound == "hbm" else 1e12), "peak": peak / (1e9 if bound == "hbm" else 1e12),
                             "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": ach / peak}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected
        # as MI355X_MICROARCH.md prescribes; collected on this same command at 256 frames, profiles/pmc_traffic.json)
        traffic, tf = None, os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf) and n == 256 and args.precision == "f32":
            pmc = json.load(open(tf))
            traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
            for k in kernels:
                kernels[k]["traffic"] = pmc.get(k, {}).get("hbm_bytes_per_launch")
        roofline = {"kernel": dom, **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac")}, "traffic": traffic}
        if dom == "k_conv_features" and args.precision == "f32":
            ex = CONV_EXECUTED_FLOP_PER_CELL * cells / (kernels[dom]["avg_ms"] * 1e-3)
            roofline["executed"] = ex / 1e12
            roofline["executed_frac"] = ex / FP32_MFMA_PEAK
            roofline["note"] = ("achieved = algorithmic FLOPs of the direct convolution (SURVEY 8d) / kernel time; the kernel computes conv2 by Winograd "
                                "F(2x2,3x3), 2.25x fewer multiplies, so the algorithmic rate can exceed the MFMA peak; executed = FLOPs actually issued "
                                "(MFMA + conv1 + transform adds)")
        res = {
            "metric": "end-to-end frames/sec (1080p->81 digits)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"configs[{1 if args.precision == 'f32' else 4}]: {n} synthetic 1080p frames per GPU per step, HIP threshold + warp + 81-cell CNN {args.precision} forward, "
                                   "generator corners (host corner search not in the timed region)",
                       "frames_per_gpu": n, "height": H, "width": W, "weights": "random-init DigitCNN (seed 1234)",
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "roofline": roofline,
            "kernels": kernels,
            "pipeline_hbm_frac": fps / world * BYTES_PER_FRAME / HBM_PEAK,
            "end_to_end_with_host_corner_search": e2e,
            "pipeline_fp32_frac": fps / world * 81 * (CONV_FLOP_PER_CELL + FC_FLOP_PER_CELL) / FP32_MFMA_PEAK,
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            m = min(n, 2 * threads)
            res["cpu_baseline"] = cpu_baseline(frames[:m].cpu().numpy(), corners[:m], sd, threads)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
