#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of the bench command -> profiles/<name>.json: per-kernel calls and average duration, stamped
with hipk_build_id() of the library that ran (bench.py quotes `rocprofv3_avg_us` next to its own event figures only when the
stamp equals the running library's).   usage: python tools/kernel_stats_to_json.py gpurun_out/<dir> profiles/<out>.json"""
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import ctypes
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk.so"))
L.hipk_build_id.restype = ctypes.c_char_p
kern = {}
for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").split("(")[0].replace(" ", "")
        if name.startswith("hipk_"):
            kern[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                          "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
json.dump({"source": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ({src})",
           "build_id": L.hipk_build_id().decode(), "kernels": kern}, open(dst, "w"), indent=1)
print(f"{len(kern)} kernels -> {dst}")
