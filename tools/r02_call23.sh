#!/bin/bash
# round 2, evidence after the two-rows-per-lane SpMV (part 1): bench line with the CPU baselines, rocprofv3 trace + PMC traffic
# passes of the bench, instruction / cycle / L2 counters of the new SpMV kernel and of the one it replaces, GMRES(30) cycle A/B
set -o pipefail
O=gpurun_out/r02c23
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 300 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log
grep -q "pytest rc=0" $O/status.txt || exit 1
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 500 bash tools/prof_bench.sh r02c23/prof_bench > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_bench $O/pmc_bench.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc bench json rc=$?" | tee -a $O/status.txt
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for v in wide pair; do
  if [ $v = pair ]; then export HIPK_SPMV_SELL_NO_WIDE=1; else unset HIPK_SPMV_SELL_NO_WIDE; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/pmc_insts_$v -- python3 $ARGS > $O/pmc_insts_$v.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_cycles_$v -- python3 $ARGS > $O/pmc_cycles_$v.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_l2_$v -- python3 $ARGS > $O/pmc_l2_$v.log 2>&1
  echo "pmc $v rc=$?" | tee -a $O/status.txt
done
unset HIPK_SPMV_SELL_NO_WIDE
python3 - $O <<'PY' > $O/spmv_counters.txt 2>&1
import csv, glob, os, sys, collections
out = sys.argv[1]
for v in ("wide", "pair"):
    for grp in ("insts", "cycles", "l2"):
        for f in sorted(glob.glob(os.path.join(out, f"pmc_{grp}_{v}", "**", "*counter_collection.csv"), recursive=True)):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                if "spmv_sell" in k:
                    print(v, k, {c: round(sum(x) / len(x), 1) for c, x in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
cat $O/spmv_counters.txt | cut -c1-400
timeout -k 10 200 python tools/gmres_variants.py 2000 HIPK_SPMV_SELL_NO_WIDE=1 "" 2>&1 | grep cycle | tee $O/variants.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02c23/bench_line.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "it/s; roofline frac", round(d["roofline"]["frac"],3), "traffic", d["roofline"].get("traffic"), "cpu", round(d["cpu_baseline"]["value"]), d["kernels"][0]["kernel"][:50], d["kernels"][0]["traffic"])
PY
du -sh $O
