#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-passes 0`.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/pmc_traffic.json

Per bench kernel name: mean raw counter per launch and the corrected HBM bytes per launch
(FETCH_SIZE is in KB and counts gfx950's 128-byte requests as 64 B -> x2 x1024; WRITE_SIZE in KB -> x1024)."""
import collections
import csv
import glob
import json
import sys

NAMES = {"k_preprocess": "k_preprocess", "k_warp_cells": "k_warp_cells", "k_conv_features": "k_conv_features", "k_fc_head": "k_fc_head"}


def means(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    kern = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        for short in NAMES:
            if short in k and "at::native" not in k:
                tot[short] += float(r["Counter_Value"])
                cnt[short] += 1
                kern[short] = k.replace("void ", "").replace("(anonymous namespace)::", "", 1).split("(")[0]
    return {k: tot[k] / cnt[k] for k in tot}, kern


fetch, kern = means(sys.argv[1], "FETCH_SIZE")
write, _ = means(sys.argv[2], "WRITE_SIZE")
out = {}
for k in NAMES:
    if k in fetch and k in write:
        out[k] = {"hbm_bytes_per_launch": int(fetch[k] * 2 * 1024 + write[k] * 1024), "FETCH_SIZE_KB_raw": fetch[k], "WRITE_SIZE_KB_raw": write[k],
                  "kernel": kern[k],
                  "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), x1024; WRITE_SIZE x1024",
                  "launch": "256 frames (20,736 cells)"}
print(json.dumps(out, indent=1))
