#!/bin/bash
# round 2, call 29: evidence on the code with the grouped walk of the two-rows-per-lane coded SpMV: whole GPU suite, walk probe per
# size, row-partitioned rehearsal shapes, bench lines (default = config 2; --scaling strong = config 5 on one GPU), rocprofv3 trace
# + PMC passes of the bench, counters of the N = 64 M SpMV (chunk walk vs groups)
set -o pipefail
O=gpurun_out/r02c29
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest_all.log
grep -q "pytest all rc=0" $O/status.txt || exit 1
timeout -k 10 400 python tools/walk_probe.py 2000 2828 4000 5657 8000 > $O/walk.log 2>&1; echo "walk rc=$?" | tee -a $O/status.txt
grep -q "walk rc=0" $O/status.txt || exit 1
for ch in 8192 16384; do
  echo "== force_ch $ch" >> $O/dist.log
  timeout -k 10 200 python tools/dist_probe.py 2000 $ch >> $O/dist.log 2>&1; echo "dist $ch rc=$?" | tee -a $O/status.txt
done
grep -E "==|dist_cg|chunk size" $O/dist.log
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --scaling strong --steps 1 --warmup 1 > $O/bench_strong_n1.json 2> $O/bench_strong_n1.err; echo "bench strong rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 timeout -k 10 300 python bench.py > $O/bench_dist_world1.json 2> $O/bench_dist_world1.err; echo "bench dist world1 rc=$?" | tee -a $O/status.txt
timeout -k 10 500 bash tools/prof_bench.sh r02c29/prof_bench > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_bench $O/pmc_bench.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc bench json rc=$?" | tee -a $O/status.txt
mkdir -p $O/n64m
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n64m/trace -- python3 tools/walk_probe.py 8000 > $O/n64m/trace.log 2>&1; echo "n64m trace rc=$?" | tee -a $O/status.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/n64m/pmc_fetch -- python3 tools/walk_probe.py 8000 > $O/n64m/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/n64m/pmc_write -- python3 tools/walk_probe.py 8000 > $O/n64m/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/n64m/pmc_l2 -- python3 tools/walk_probe.py 8000 > $O/n64m/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/n64m/pmc_sq -- python3 tools/walk_probe.py 8000 > $O/n64m/pmc_sq.log 2>&1; echo "n64m pmc rc=$?" | tee -a $O/status.txt
python3 tools/pmc_kernels.py $O/n64m hipk_spmv_sell hipk_cg_ > $O/n64m/pmc_summary.txt 2>&1
f=$(find $O/n64m/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" > $O/n64m/kernel_stats.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
for f in ("bench_line", "bench_strong_n1", "bench_dist_world1"):
    try:
        d = json.loads(open(f"gpurun_out/r02c29/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d["unit"][:30], "ms/step", round(d["ms_per_step"], 1), "roofline", round(d["roofline"]["frac"], 3), d["roofline"].get("kernel", "")[:60])
    except Exception as e:
        print(f, "unreadable:", e)
PY
du -sh $O
