#!/usr/bin/env python3
"""BASELINE.json configs[2]: one camera feed at 30 fps, per-frame latency of the hot path on one MI355X.

Per frame:  pinned host frame --H2D--> K1 (bit image) -> despeckle --D2H bits--> host contour corner search --> Minv --H2D--> K2 -> K3
            --D2H--> 81 digits.   The two device segments (K1; K2->K3) are hipGraph-captured once and replayed.
Reports p50/p90/p99 of (a) the whole frame -> digits latency as a host clock around it, (b) its device segments.
Prints one JSON line.  Not the headline metric (that is bench.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--fps", type=float, default=30.0, help="feed rate; 0 = back to back")
    ap.add_argument("--glue", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import sudoku_vision_amd as sva
    from sudoku_vision_amd.synth import random_state_dict, synth_frames

    torch.cuda.set_device(0)
    ctx = sva.default_context()
    H, W = 1080, 1920
    pool, corners_gt, _ = synth_frames(32, H, W, seed=77, device="cuda")
    host_pool = [pool[i].cpu().pin_memory() for i in range(32)]
    ctx.load_state_dict(random_state_dict(1234))
    ctx.reserve(81)

    frame_d = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
    binary_h = torch.empty((1, H, W // 32), dtype=torch.int32).pin_memory()        # 1 bit per pixel over PCIe
    bits_d = torch.empty((1, H, W // 32), dtype=torch.int32, device="cuda")
    minv_h = torch.empty((1, 9), dtype=torch.float64).pin_memory()
    minv_d = torch.empty((1, 9), dtype=torch.float64, device="cuda")
    out = {"logits": torch.empty((1, 81, 10), device="cuda"), "digits": torch.empty((1, 81), dtype=torch.uint8, device="cuda"),
           "conf": torch.empty((1, 81), device="cuda")}
    digits_h = torch.empty((1, 81), dtype=torch.uint8).pin_memory()
    binary_d = torch.empty((1, H, W), dtype=torch.uint8, device="cuda")

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        # warm-up outside capture, then capture the two device segments
        b = ctx.preprocess(frame_d)
        minv_d.copy_(torch.from_numpy(sva.Context.corners_to_minv(corners_gt[:1]).reshape(1, 9)))
        ctx.frames_to_digits(frame_d, minv_d, out=out, glue=args.glue)
        stream.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=stream):
            ctx.despeckle_bits(ctx.preprocess_bits(frame_d, out=bits_d))     # K1 writes the bit image, the speck filter works on it in place
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=stream):
            ctx.frames_to_digits(frame_d, minv_d, out=out, glue=args.glue)

    lat, seg1, seg2, hostms, found = [], [], [], [], 0
    period = 1.0 / args.fps if args.fps > 0 else 0.0
    t_next = time.perf_counter()
    for i in range(args.frames):
        if period:
            while time.perf_counter() < t_next:
                pass
            t_next += period
        src = host_pool[i % 32]
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            frame_d[0].copy_(src, non_blocking=True)
            g1.replay()
            binary_h.copy_(bits_d, non_blocking=True)
            stream.synchronize()
            t1 = time.perf_counter()
            cc, ff = sva.host.find_grid_corners_bits_batch(binary_h.numpy(), H, W, threads=1)
            c = cc[0] if ff[0] else None
            t2 = time.perf_counter()
            if c is not None:
                found += 1
                minv_h.copy_(torch.from_numpy(sva.Context.corners_to_minv(c[None].astype(np.float32)).reshape(1, 9)))
                minv_d.copy_(minv_h, non_blocking=True)
                g2.replay()
                digits_h.copy_(out["digits"], non_blocking=True)
            stream.synchronize()
        t3 = time.perf_counter()
        lat.append((t3 - t0) * 1e3); seg1.append((t1 - t0) * 1e3); hostms.append((t2 - t1) * 1e3); seg2.append((t3 - t2) * 1e3)

    def pct(v):
        v = np.sort(np.array(v[10:]))
        return {"p50": float(np.percentile(v, 50)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99))}

    print(json.dumps({"metric": "per-frame latency, 1080p frame -> 81 digits (configs[2]: streamed feed, hipGraph replay)", "unit": "ms",
                      "fps_feed": args.fps, "frames": args.frames, "grids_found": found,
                      "frame_to_digits": pct(lat), "h2d_frame+K1_graph+d2h_binary": pct(seg1), "host_corner_search_1_thread": pct(hostms),
                      "h2d_minv+K2K3_graph+d2h_digits": pct(seg2), "glue": args.glue}))


if __name__ == "__main__":
    main()
