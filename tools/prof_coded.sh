#!/bin/bash
# PMC passes (each its own run, no trace domains) for the coded SpMV kernel, stand-alone probe.
OUT=gpurun_out/${1:-prof_coded}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 tools/spmv_f32_probe.py"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_inst -- $CMD > $OUT/pmc_inst.log 2>&1
# (a pass with TA_* counters hung on this pool and was killed by the harness: TA/TCP counters are not requested)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
python3 - $OUT <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "spmv" in k:
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
cat $OUT/summary.txt
grep -il "error\|invalid\|not found" $OUT/*.log | head
