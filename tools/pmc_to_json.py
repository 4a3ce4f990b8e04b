#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from the rocprofv3 --pmc passes of tools/prof_bench.sh.
usage: python tools/pmc_to_json.py gpurun_out/<name> profiles/<out>.json
Launches that returned at once (iterations past the stop word) are dropped: only launches whose counter is at
least half of the kernel's maximum enter the mean.  traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): gfx950's
FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
KEYS = {"hipk_spmv_sell_loop_kernel<double": "spmv", "hipk_spmv_kernel<double": "spmv_plain",
        "hipk_cg_update_kernel<double": "cg_update", "hipk_cg_direction_kernel<double": "cg_direction"}


def means(counter_dir, counter):
    out = {}
    for f in glob.glob(os.path.join(src, counter_dir, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for name, v in acc.items():
            for pat, key in KEYS.items():
                if pat in name:
                    top = max(v)
                    real = [a for a in v if a >= 0.5 * top]
                    out[key] = {"kernel": name.split("(")[0].replace("void ", ""), "mean_KB": sum(real) / len(real),
                                "launches": len(real), "dropped_noop_launches": len(v) - len(real)}
    return out


fetch, write = means("pmc_fetch", "FETCH_SIZE"), means("pmc_write", "WRITE_SIZE")
kern = {}
for key in fetch:
    w = write.get(key, {"mean_KB": 0.0})
    kern[key] = {"kernel": fetch[key]["kernel"], "FETCH_SIZE_KB_mean": fetch[key]["mean_KB"],
                 "WRITE_SIZE_KB_mean": w["mean_KB"], "launches": fetch[key]["launches"],
                 "dropped_noop_launches": fetch[key]["dropped_noop_launches"],
                 "traffic_bytes_per_launch": int(round((2.0 * fetch[key]["mean_KB"] + w["mean_KB"]) * 1000.0))}
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/prof_bench.sh ({src})",
           "fetch_correction": 2.0,
           "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE; counters are KB (1000 B); FETCH_SIZE includes Infinity-Cache hits",
           "kernels": kern}, open(dst, "w"), indent=1)
print(json.dumps(kern, indent=1))
