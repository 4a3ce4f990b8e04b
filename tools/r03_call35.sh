#!/bin/bash
# instruction counters of the one-launch CG loop (one --pmc pass, kernel trace only): VALU / LDS / scalar / vector-memory instructions per launch
mkdir -p gpurun_out/r03c35
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/r03c35/pmc -- python3 $R/tools/cg_mid_probe.py mid > $R/gpurun_out/r03c35/probe.log 2>&1); echo "pmc rc=$?"
python3 - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03c35/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hipk_cg_mid_kernel" in r["Kernel_Name"]:
            rows[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (k, grid), c in sorted(rows.items(), key=lambda x: int(x[0][1])):
    out.append({"kernel": k, "grid_threads": int(grid), "launches": len(next(iter(c.values()))), **{n: sum(v) / len(v) for n, v in c.items()}})
json.dump(out, open("gpurun_out/r03c35/cg_mid_inst_counters.json", "w"), indent=1)
for o in out: print(o)
PY
grep '"mid"' gpurun_out/r03c35/probe.log | awk 'NR%2==0' | cut -c1-140
rm -rf gpurun_out/r03c35/pmc
