#!/usr/bin/env python3
"""Throughput of the JPEG front end (scope row N4) on synthetic 1080p frames encoded with Pillow (quality 90, 4:2:0):
this build (host Huffman threads + HIP reconstruction) next to Pillow's libjpeg-turbo decode on the same host threads.
Prints one JSON line; every decoded frame is checked bit-for-bit against Pillow's."""
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sudoku_vision_amd as sva  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    threads = min(16, os.cpu_count() or 1)
    ctx = sva.default_context()
    frames, _, _ = synth_frames(n, 1080, 1920, seed=77, device="cuda")
    host = frames.cpu().numpy()
    datas = []
    for i in range(n):
        b = io.BytesIO()
        Image.fromarray(host[i][..., ::-1].copy()).save(b, "JPEG", quality=90, subsampling=2)
        datas.append(b.getvalue())
    mean_kb = sum(len(d) for d in datas) / n / 1e3

    def pil(d):
        return np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))

    with ThreadPoolExecutor(threads) as ex:
        want = list(ex.map(pil, datas))
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            list(ex.map(pil, datas))
        pil_fps = n * reps / (time.perf_counter() - t0)

    res = {}
    exact = True
    for mode, dense in (("sparse", False), ("dense", True)):
        for _ in range(2):                                   # warm-up: both pinned staging sets, worker threads
            out = ctx.imdecode_batch(datas, threads, dense=dense)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        exact = exact and all((got[i][..., ::-1] == want[i]).all() for i in range(n))
        t0 = time.perf_counter()
        for _ in range(2 * reps):
            ctx.imdecode_batch(datas, threads, dense=dense)
        torch.cuda.synchronize()
        res[mode] = {"fps": n * 2 * reps / (time.perf_counter() - t0), "pcie_bytes_per_frame": ctx._jpeg_last_bytes}
    fps = res["sparse"]["fps"]

    # the device half alone
    from sudoku_vision_amd import host as svh
    info, coef, quant = svh.jpeg_entropy_decode(datas[0])
    dcoef = torch.from_numpy(np.concatenate([coef, np.zeros((-len(coef)) % 8, np.int16), quant.view(np.int16).ravel()])).cuda()
    qoff = (len(coef) + 7) // 8 * 8
    frame = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
    import ctypes as C
    lib = sva._native.lib()
    call = lambda: lib.sv_jpeg_reconstruct_bgr_u8(ctx._h, C.byref(info), C.c_void_p(dcoef.data_ptr()), C.c_void_p(dcoef.data_ptr() + 2 * qoff),
                                                  C.c_void_p(frame.data_ptr()), 1920 * 3, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / 50
    t0 = time.perf_counter()
    for d in datas[:16]:
        svh.jpeg_entropy_decode(d, coef=coef, quant=quant)
    huff_ms = (time.perf_counter() - t0) / 16 * 1e3
    alg_bytes = info.coef_count * 2 + info.coef_count + info.coef_count + 1080 * 1920 * 3     # coef in, planes out+in, frame out
    print(json.dumps({"metric": "JPEG front end, 1080p 4:2:0 q90 frames/s", "frames": n, "mean_file_kB": mean_kb, "host_threads": threads,
                      "this_build_fps": fps, "transport": res, "pillow_libjpeg_turbo_fps_same_threads": pil_fps, "bit_exact_vs_pillow": bool(exact),
                      "huffman_ms_per_frame_one_thread": huff_ms, "device_reconstruct_ms_per_frame": dev_ms,
                      "device_reconstruct_GBps": alg_bytes / dev_ms / 1e6}))


if __name__ == "__main__":
    main()
