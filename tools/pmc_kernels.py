#!/usr/bin/env python3
"""Per-kernel means of every counter found under a directory of rocprofv3 --pmc passes (any command), launches below half of
the kernel's maximum dropped (no-op launches past a stop word); traffic = 2 x FETCH_SIZE + WRITE_SIZE in MB where both exist
(gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md).
usage: python tools/pmc_kernels.py gpurun_out/<dir> [name-substring ...]"""
import collections
import csv
import glob
import os
import sys

src, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pats and not any(p in k for p in pats):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    row = {}
    for c, v in acc[k].items():
        big = [x for x in v if x >= 0.5 * max(v)] or v
        row[c] = sum(big) / len(big)
    n = len(next(iter(acc[k].values())))
    out = {c: round(x, 1) for c, x in row.items()}
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        out["traffic_MB"] = round((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in row and "TCC_MISS_sum" in row:
        out["l2_hit_frac"] = round(row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 3)
    print(k[:90], "launches", n, out)
