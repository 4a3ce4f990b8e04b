#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-kernel mean of each PMC counter)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pat):
    return sorted(glob.glob(os.path.join(out, "**", pat), recursive=True))
for f in find("*kernel_stats.csv"):
    print("== kernel stats:", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"  {r.get('Name','')[:70]:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
for f in find("*counter_collection.csv"):
    print("== counters:", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "spmv" in k or "cg_" in k or "dot" in k or "axpy" in k:
            print("  ", k, {c: (sum(v) / len(v), len(v)) for c, v in cs.items()})
