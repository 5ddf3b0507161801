#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) over `python3 bench.py --no-cpu-baseline --no-e2e` (the bench's default step count), and optionally the same two passes with `--precision bf16`.

    python tools/pmc_traffic.py <FETCH_SIZE dir> <WRITE_SIZE dir> [<bf16 FETCH_SIZE dir> <bf16 WRITE_SIZE dir>] > profiles/pmc_traffic.json

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


def section(fetch_dir, write_dir):
    fetch, kern = means(fetch_dir, "FETCH_SIZE")
    write, _ = means(write_dir, "WRITE_SIZE")
    out = {}
    for k in NAMES:
        if k in fetch and k in write:
            out[k] = {"hbm_bytes_per_launch": int(fetch[k] * 2 * 1024 + write[k] * 1024), "FETCH_SIZE_KB_raw": fetch[k], "WRITE_SIZE_KB_raw": write[k],
                      "kernel": kern[k],
                      "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), x1024; WRITE_SIZE x1024",
                      "launch": "256 frames (20,736 cells)"}
    return out


import datetime
out = section(sys.argv[1], sys.argv[2])
out["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) over `python3 bench.py --no-cpu-baseline --no-e2e`, 256 frames per launch, "
                  f"collected {datetime.date.today().isoformat()} (tools/run_round_profile.sh)")
if len(sys.argv) > 4:
    out["bf16"] = section(sys.argv[3], sys.argv[4])
print(json.dumps(out, indent=1))
