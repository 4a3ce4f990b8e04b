#!/bin/bash
# round 2, call 43: whole GPU suite, bench line (default) and config 5 on one device on the final code (flat direction step at N > 32 M)
set -o pipefail
O=gpurun_out/r02c43
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest_all.log | cut -c1-300
grep -q "pytest all rc=0" $O/status.txt || exit 1
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --scaling strong --steps 1 --warmup 1 > $O/bench_strong_n1.json 2> $O/bench_strong_n1.err; echo "bench strong rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/dist_probe.py 1000 32768 8000 2000 > $O/dist_c5.log 2>&1; echo "dist c5 rc=$?" | tee -a $O/status.txt
grep -E "dist_cg|chunk size" $O/dist_c5.log
python - <<'PY'
import json
for f in ("bench_line", "bench_strong_n1"):
    d = json.loads(open(f"gpurun_out/r02c43/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), "it/s", "roofline", round(d["roofline"]["frac"], 3), [(k["key"], round(k["avg_launch_us"], 1), round(k.get("frac_of_hbm_peak") or 0, 3)) for k in d["kernels"]])
PY
