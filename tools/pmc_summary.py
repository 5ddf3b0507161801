#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean of each counter (this repo's kernels only)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "anonymous namespace" not in k or "at::native" in k:
        continue
    name = k.split("::")[1].split("(")[0][:40]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(name, r["Counter_Name"])] += 1
for n, d in agg.items():
    print(n, {c: round(v / cnt[(n, c)]) for c, v in sorted(d.items())})
