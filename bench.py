#!/usr/bin/env python3
"""Benchmark of the frame -> digits hot path on MI355X (BASELINE.json metric: frames/s, 1080p -> 81 digits).

    python bench.py --gpus N --steps K --warmup W            # N > 1 without a launcher: bench.py starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic 1080p frames already resident in HBM.  The line carries two
measurements of the same batch:
  value (value_kind "end_to_end", BASELINE.json's metric): K1 (frame -> bit image) -> speck filter -> D2H -> host contour corner
      search -> K2 (frame + homography -> 81 cells) -> K3 (cells -> logits, digits), software-pipelined over chunks of the batch
      (sudoku-vision_amd/pipeline.py FramePipeline).  Three timed regions of exactly K steps each, every one bracketed by
      barrier + synchronize, MAX over ranks; value and ms_per_step are the median region's and all three are listed.
  value_device_only + roofline + kernels: K1 + K2 + K3 with the generator's corners (no host work in the timed region), K steps,
      every hot kernel bracketed by HIP events on its launch stream inside the library.

Workloads
  configs1 (default)  BASELINE.json configs[1] per GPU: 256 synthetic 1080p frames per step, f32-grade CNN.
                      Weak scaling: every rank owns its own 256-frame pool.
  configs3            BASELINE.json configs[3]: 100,000 frames dealt round-robin (frame i -> rank i mod N, sharding.shard_indices),
                      each rank cycling its 256-frame pool; a step = one pass over the 100,000 frames.  Strong scaling.
  --precision bf16    BASELINE.json configs[4]: conv2 / fc1 on bf16 MFMA (digit-index parity only).

Ranks never exchange data: the process group (gloo, host side -- no RCCL dependency) only lines the ranks
up for timing and takes the MAX of their elapsed times.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per unit (restated in DESIGN.md section 4)
BYTES_PER_FRAME = 9_916_344          # read 6,220,800 + <=1,555,200 ; write 2,073,600 + 63,504 + 3,240
K1_BYTES_PER_FRAME = 6_220_800 + 2_073_600
K2_BYTES_PER_FRAME = 1_555_200 + 63_504
CONV_FLOP_PER_CELL = 451_584 + 7_225_344          # direct 3x3 convolutions (conv1 + conv2), MAC = 2 FLOP
FC_FLOP_PER_CELL = 802_816 + 2_560
HBM_PEAK = 8.0e12                    # B/s, MI355X_MICROARCH.md
FP32_MFMA_PEAK = 157.3e12            # FLOP/s, v_mfma_f32_* (= fp32 vector peak)
BF16_MFMA_PEAK = 2.5e15              # FLOP/s dense, v_mfma_f32_*_bf16
F16_MFMA_PEAK = 2.5e15               # FLOP/s dense, v_mfma_f32_*_f16 (same rate as bf16)


def issued_flop_per_cell(info):
    """FLOPs the CNN kernels ISSUE per cell (what their roofline fractions are priced on) -> (conv, fc, peak FLOP/s of the pipe).
    Default pair (algo 4): v_mfma_f32_16x16x32_f16 instructions (16384 FLOP each; three partial products per f32-grade product,
    conv1's K padded 16 -> 32) against the dense f16 MFMA peak.  f32-MFMA kernels (what out-of-range weights run on; the test-only library's
    Winograd forms too): conv2 direct (3600 v_mfma_f32_16x16x4_f32 of 2048 FLOP per cell) or by Winograd F(2x2,3x3) (1568), conv1 and the
    Winograd transforms' adds on the VALU, fc1's 802,816 FLOP per cell, against the f32 MFMA peak."""
    if info["mfma_f16_conv"]:
        return info["mfma_f16_conv"] * 16384, info["mfma_f16_fc"] * 16384, F16_MFMA_PEAK
    conv = (info["mfma_conv2"] + info["mfma_conv1"]) * 2048
    if not info["mfma_conv1"]:
        conv += 451_584
    if info["algo"] == 2:
        conv += 49 * 32 * 32 + 49 * 64 * 24
    return conv, FC_FLOP_PER_CELL, FP32_MFMA_PEAK


def cpu_baseline_test_image(sd, seconds=4.0):
    """BASELINE configs[0] -- pipeline/run.py:244-355 on one data/test_images frame -- with the CPU port in the reference's place
    (cv2 and the reference's Python do not travel to the GPU box): the committed tests/golden/sample_4.jpg (the photo the reference's
    own integration test uses, tests/test_integration.py:121), decoded once with Pillow outside the timed region, then per pass
    K1 -> contour corner search -> warp + 81 cells -> preprocess_cell (CLAHE + threshold) -> DigitCNN, on one thread as run.py runs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    from PIL import Image
    import cnn_oracle
    import sv_oracle
    path = os.path.join(ROOT, "tests", "golden", "sample_4.jpg")
    if not os.path.exists(path):
        return None
    img = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])
    torch.set_num_threads(1)
    t0, passes, found = time.perf_counter(), 0, False
    while True:
        binary = sv_oracle.preprocess_for_grid_detection(img)
        quad = sv_oracle.find_grid_contour(binary)
        found = quad is not None
        if found:
            cells = sv_oracle.warp_cells(img, np.asarray(quad, np.float32).reshape(4, 2))
            cnn_oracle.predict(sd, sv_oracle.cells_to_input(sv_oracle.preprocess_cells(cells))[:, None])
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= 50:
            break
    return {"value": passes / dt, "unit": "frames/s", "ms_per_frame": dt / passes * 1e3, "cores": 1, "kind": "port", "grid_found": found,
            "sample": f"{passes} pass(es) over tests/golden/sample_4.jpg ({img.shape[0]}x{img.shape[1]}, = the reference's data/test_images/sample_4.jpg) in "
                      f"{dt:.1f} s: C oracle K1 + corner search + K2 + preprocess_cell, torch-CPU DigitCNN, one thread (BASELINE configs[0])"}


def cpu_baseline(frames_host, corners, sd, threads):
    """The oracle (CPU port of the reference arithmetic) on a bounded sample: K1+K2 in C, one frame per
    thread; CNN with torch-CPU on all cells.  Baseline only.  The one place in this file that touches oracle/."""
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


def rank_identity(rank, local_rank, device_index):
    """What the driver needs to see that N ranks sat on N distinct GPUs: torch's device name, PCI address and UUID of this rank's GPU."""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    pci = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return {"rank": rank, "local_rank": local_rank, "device_index": device_index, "device": p.name, "pci_bus_id": pci,
            "uuid": str(getattr(p, "uuid", "")), "pid": os.getpid()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 100 (configs1: 256 frames each) / 1 (configs3: 100,000 frames each)")
    ap.add_argument("--warmup", type=int, default=None, help="default 20 (configs1) / 1 (configs3)")
    ap.add_argument("--frames", type=int, default=256, help="frames in each GPU's resident pool (= frames per GPU per step in configs1)")
    ap.add_argument("--workload", choices=["configs1", "configs3"], default="configs1")
    ap.add_argument("--total-frames", type=int, default=100_000, help="configs3: frames dealt round-robin over the ranks per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (headline); bf16 = configs[4]: conv2/fc1 on bf16 MFMA, digit-index parity only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end measurement: value = the device-only figure, value_kind says so")
    ap.add_argument("--e2e-passes", type=int, default=None, help="deprecated: 0 = --no-e2e")
    ap.add_argument("--chunk", type=int, default=None, help="end-to-end pipeline: frames per chunk (default: largest of 256, 128, 64, ... dividing --frames)")
    ap.add_argument("--depth", type=int, default=None, help="end-to-end pipeline: chunks in flight")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100 if args.workload == "configs1" else 1
    if args.warmup is None:
        args.warmup = 20 if args.workload == "configs1" else 1
    if args.e2e_passes == 0:
        args.no_e2e = True

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                                       # no launcher: be the launcher (before anything touches the GPU)
            from sudoku_vision_amd.sharding import launch_local_ranks
            raise SystemExit(launch_local_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to benchmark a different GPU count than asked for")

    import numpy as np
    import torch

    import sudoku_vision_amd as sva
    from sudoku_vision_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    # SV_BENCH_REHEARSE=1: rehearse the multi-rank code path on a one-GPU box (all ranks on cuda:0)
    rehearse = os.environ.get("SV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (SV_BENCH_REHEARSE=1 puts every rank on cuda:0)")
    torch.cuda.set_device(local_rank)
    sharding.init()                                             # gloo; only for the timing barrier, the MAX of elapsed and the rank identities
    ranks = sharding.gather_objects(rank_identity(rank, local_rank, torch.cuda.current_device()))

    from sudoku_vision_amd.synth import random_state_dict, synth_frames

    ctx = sva.default_context()
    n = args.frames
    H, W = 1080, 1920
    frames, corners, _ = synth_frames(n, H, W, seed=1234 + rank, device="cuda")
    sd = random_state_dict(1234)                                # random-init weights of the DigitCNN architecture
    ctx.load_state_dict(sd)
    ctx.reserve(n * 81)
    if args.precision == "bf16":
        ctx.set_precision(ctx.PREC_BF16)
    minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners))
    out = {"logits": torch.empty((n, 81, 10), dtype=torch.float32, device="cuda"),
           "digits": torch.empty((n, 81), dtype=torch.uint8, device="cuda"),
           "conf": torch.empty((n, 81), dtype=torch.float32, device="cuda")}
    binary = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")

    def batch(m):
        """the device-only hot path over the first m frames of the pool"""
        ctx.preprocess(frames[:m], out=binary[:m])                                   # K1: what the host corner search consumes
        ctx.frames_to_digits(frames[:m], minv[:m], out={k: v[:m] for k, v in out.items()})   # K2 -> K3

    if args.workload == "configs1":
        frames_per_step_rank, frames_per_step_job = n, n * world
    else:
        # frame i of the 100,000 goes to rank i mod world; the k-th frame a rank owns is its pool frame k mod n
        frames_per_step_rank, frames_per_step_job = len(sharding.shard_indices(args.total_frames, rank, world)), args.total_frames
    mine = frames_per_step_rank

    def step():
        for _ in range(mine // n):
            batch(n)
        if mine % n:
            batch(mine % n)

    def barrier():
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()

    # ---- device-only: the kernels' own rates (roofline) ------------------------------------------------------------------------------
    # clock ramp: an idle MI355X needs some tens of milliseconds of load before it runs at its sustained clock (12 steps from idle run
    # ~13 % slower than the same steps after 100 ms of load) -- untimed steps of the same work, before and apart from the W warm-up steps
    preroll = 0
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.1:
        batch(n)
        torch.cuda.synchronize()
        preroll += 1
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_dev = time.perf_counter() - t0
    per_kernel = ctx.timing_end()
    elapsed_dev = sharding.max_over_ranks(elapsed_dev)

    # ---- end to end (BASELINE's metric): the same steps with the host corner search in the loop -----------------------------------
    e2e = None
    if not args.no_e2e:
        from sudoku_vision_amd.pipeline import FramePipeline, host_cpu_budget
        # the box's CPU share is 16 cores per GPU, enforced as a cgroup quota: stay two under it (this thread + the HIP runtime's)
        # (the ranks of one node are children of one launcher: they share its cgroup, hence its quota, and the host's CPUs)
        budget = host_cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        host_threads = max(1, min(16, budget) - 2)
        chunk = args.chunk or next(c for c in (256, 128, 64, 32, 16, 8, 4, 2, 1) if n % c == 0)
        kw = {"depth": args.depth} if args.depth else {}
        pipe = FramePipeline(ctx, H, W, chunk=chunk, host_threads=host_threads, **kw)
        # warm-up: W steps, and at least 1 s of them (page-locks, worker threads and their scratch, host caches, GPU and CPU clocks after the
        # idle stretch in which the pipeline's buffers were allocated: regions started 0.3 s into the streaming still climbed, e.g. 144 k ->
        # 168 k -> 193 k frames/s on one box; profiles/r03_e_bench.json)
        t_w = time.perf_counter()
        pipe.run(frames, out=out, total=max(n, min(mine, n) * args.warmup))
        while time.perf_counter() - t_w < 1.0 and args.warmup > 0:
            pipe.run(frames, out=out, total=max(n, min(mine, 8 * n)))
        # three timed regions of exactly K steps; value = their median and all three are listed: on a shared host a region now and then
        # contains a 40-60 ms stall of one search call (tools/dev/search_outliers.py: with or without GPU work in flight), which says
        # nothing about the pipeline
        regions = []
        for _ in range(3):
            barrier()
            t1 = time.perf_counter()
            res_e2e = pipe.run(frames, out=out, total=mine * args.steps)     # K steps streamed back to back, pipeline kept full
            barrier()
            regions.append(sharding.max_over_ranks(time.perf_counter() - t1))
        dt = sorted(regions)[1]
        seen = min(n, mine * args.steps)
        err = np.abs(res_e2e["corners"][:seen].astype(np.float32)[:, :, None, :] - corners[:seen, None, :, :]).sum(-1).min(-1).max()
        e2e = {"value": frames_per_step_job * args.steps / dt, "unit": "frames/s", "ms_per_step": dt / args.steps * 1e3, "host_threads_per_gpu": host_threads,
               "regions": [frames_per_step_job * args.steps / t for t in regions], "aggregate": "median of three timed regions of `steps` steps each",
               "grids_found": int(res_e2e["found"][:seen].sum()), "of": seen, "max_corner_error_px": float(err),
               "dense_fallbacks": pipe.dense_fallbacks, "note": pipe.describe()}

    if rank == 0:
        total_frames = frames_per_step_job * args.steps
        fps_dev = total_frames / elapsed_dev
        launched = frames_per_step_rank * args.steps               # frames rank 0 pushed through each kernel in the timed region
        conv_info = ctx.conv_kernel_info()
        per_frame = {"k_preprocess": ("hbm", K1_BYTES_PER_FRAME), "k_warp_cells": ("hbm", K2_BYTES_PER_FRAME),
                     "k_conv_features": ("mfma", CONV_FLOP_PER_CELL * 81), "k_fc_head": ("mfma", FC_FLOP_PER_CELL * 81)}
        kernels = {}
        for name, (ms, cnt) in per_kernel.items():
            if not cnt or name not in per_frame:                    # a kernel id this script has no price for (e.g. the fused launch) is not on this path
                continue
            bound, units = per_frame[name]
            avg = ms / cnt * 1e-3                                   # average launch duration (HIP events on the launch stream)
            per_launch = units * launched / cnt                     # algorithmic bytes / FLOPs of an average launch
            peak = HBM_PEAK if bound == "hbm" else (BF16_MFMA_PEAK if args.precision == "bf16" else FP32_MFMA_PEAK)
            ach = per_launch / avg
            scale = 1e9 if bound == "hbm" else 1e12
            kernels[name] = {"bound": bound, "avg_ms": ms / cnt, "launches": cnt, "achieved": ach / scale, "peak": peak / scale,
                             "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": ach / peak}
        if args.precision == "f32":
            # `frac` of the CNN kernels is priced on the FLOPs they ISSUE on the pipe they issue them on; the direct f32 convolution /
            # matrix product the algorithmic figure counts is kept beside it as the f32-equivalent rate
            conv_f, fc_f, pipe_peak = issued_flop_per_cell(conv_info)
            for name, per_cell in (("k_conv_features", conv_f), ("k_fc_head", fc_f)):
                if name not in kernels:
                    continue
                k = kernels[name]
                k["algorithmic_equiv"], k["algorithmic_equiv_frac_of_f32_peak"] = k["achieved"], k["frac"]
                issued = per_cell * 81 * launched / k["launches"] / (k["avg_ms"] * 1e-3)
                k["achieved"], k["peak"], k["frac"] = issued / 1e12, pipe_peak / 1e12, issued / pipe_peak
                k["issued_flop_per_cell"], k["algo"] = per_cell, conv_info["name"]
                k["pipe"] = "v_mfma_f32_16x16x32_f16" if conv_info["mfma_f16_conv"] else "v_mfma_f32_16x16x4_f32"
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
        # MI355X_MICROARCH.md prescribes).  NOT measured in this run: read from profiles/pmc_traffic.json, which tools/pmc_traffic.py
        # writes from a rocprofv3 run of this same command at 256 frames -- traffic_source says which
        traffic, traffic_source, tf = None, None, os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf) and n == 256 and args.workload == "configs1":
            pmc = json.load(open(tf))
            sect = pmc if args.precision == "f32" else pmc.get("bf16", {})
            traffic = sect.get(dom, {}).get("hbm_bytes_per_launch")
            for k in kernels:
                kernels[k]["traffic"] = sect.get(k, {}).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/pmc_traffic.json: " + str(pmc.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py` at 256 frames"))
                              + " -- a constant from that profile, not measured by this run")
        roofline = {"kernel": dom, **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac")}, "traffic": traffic,
                    "traffic_source": traffic_source, "measured_in": "device-only steps of this run (HIP events on the launch stream)"}
        if "algorithmic_equiv" in kernels[dom]:
            roofline.update({k: kernels[dom][k] for k in ("algorithmic_equiv", "algorithmic_equiv_frac_of_f32_peak", "issued_flop_per_cell", "algo", "pipe")})
            roofline["note"] = ("achieved/frac = FLOPs the kernel issues on its matrix pipe / kernel time / that pipe's dense peak; algorithmic_equiv = the direct "
                                "f32 convolution / matrix product's FLOPs (SURVEY 8d) / kernel time")
        cfg_id = 3 if args.workload == "configs3" else (1 if args.precision == "f32" else 4)
        if args.workload == "configs3":
            wl = (f"configs[3]: {args.total_frames} synthetic 1080p frames per step dealt round-robin over {world} GPU(s) (frame i -> rank i mod N), "
                  f"each rank cycling its resident {n}-frame pool in {n}-frame batches")
        else:
            wl = f"configs[{cfg_id}]: {n} synthetic 1080p frames per GPU per step"
        wl += f", HIP threshold + warp + 81-cell CNN {args.precision} forward"
        wl += ("; value = end to end (K1 -> D2H -> host contour corner search -> K2 -> K3, pipelined); value_device_only = K1 + K2 + K3 with the generator's corners"
               if e2e else "; device-only: generator corners, the host corner search is not in the timed region")
        fps = e2e["value"] if e2e else fps_dev
        res = {
            "metric": "end-to-end frames/sec (1080p->81 digits)" if e2e else "frames/sec (1080p->81 digits), device-only",
            "value": fps, "value_kind": "end_to_end" if e2e else "device_only",
            "value_device_only": fps_dev, "value_end_to_end": e2e["value"] if e2e else None, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_steps_before_warmup": preroll,
            "ms_per_step": e2e["ms_per_step"] if e2e else elapsed_dev / args.steps * 1e3, "ms_per_step_device_only": elapsed_dev / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.workload == "configs3" else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "arithmetic": (conv_info["name"] + "; logits within 1e-4 of the PyTorch-CPU f32 model (measured ~1e-6), digits equal"
                           if args.precision == "f32" else "bf16 operands, f32 accumulation; digit-index parity only"),
            "config": {"workload": wl, "frames_per_gpu": n if args.workload == "configs1" else frames_per_step_rank,
                       "frames_per_step": frames_per_step_job, "height": H, "width": W, "weights": "random-init DigitCNN (seed 1234)",
                       "parallelism": f"frames sharded over {world} GPU(s), one process per GPU, no data-path collective (gloo barrier for timing only)"},
            "per_gpu_value": fps / world, "per_gpu_value_device_only": fps_dev / world,
            "ranks": ranks,
            "roofline": roofline,
            "kernels": kernels,
            "pipeline_hbm_frac": fps_dev / world * BYTES_PER_FRAME / HBM_PEAK,
            "end_to_end_with_host_corner_search": e2e,
            # whole-step view of the CNN's ALGORITHMIC (f32-equivalent) FLOPs against the f32 MFMA peak -- a context figure, not a roofline
            # fraction: the default kernels do this arithmetic on the f16 pipe (their own fractions are in "kernels", priced on issued FLOPs)
            "pipeline_algorithmic_cnn_flops_vs_f32_mfma_peak": fps_dev / world * 81 * (CONV_FLOP_PER_CELL + FC_FLOP_PER_CELL) / FP32_MFMA_PEAK,
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            m = min(n, 2 * threads)
            res["cpu_baseline"] = cpu_baseline(frames[:m].cpu().numpy(), corners[:m], sd, threads)
            res["cpu_baseline_test_image"] = cpu_baseline_test_image(sd)
        print(json.dumps(res), flush=True)
    sharding.shutdown()


if __name__ == "__main__":
    main()
