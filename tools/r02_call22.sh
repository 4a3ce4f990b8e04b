#!/bin/bash
# round 2: after the two-rows-per-lane SpMV -- full GPU suite, bench line, config 5 at N = 1 (N = 64 M, chunks of 128 tiles),
# GMRES(30) cycle at N = 4 M, configs 2-4
set -o pipefail
O=gpurun_out/r02c22
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_gpu.log
grep -q "pytest gpu rc=0" $O/status.txt || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --scaling strong --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_strong.json 2> $O/bench_strong.err; echo "bench strong rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/gmres_variants.py 2000 HIPK_SPMV_SELL_NO_WIDE=1 "" 2>&1 | grep cycle | tee $O/variants.log
timeout -k 10 300 python tools/bench_solvers.py 2000 2>/dev/null | grep "^{" > $O/solvers.jsonl; echo "solvers rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_line", "bench_strong"):
    d=json.loads(open(f"gpurun_out/r02c22/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), "it/s; roofline frac", round(d["roofline"]["frac"],3), d["spmv"]["in_loop_us"], d["kernels"][0]["kernel"][:60], d["config"].get("info"))
PY
cut -c1-260 $O/solvers.jsonl
