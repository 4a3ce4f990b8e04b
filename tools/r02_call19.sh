#!/bin/bash
set -o pipefail
O=gpurun_out/r02c19
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
HIPK_BENCH_DIST=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist.json 2> $O/bench_dist.err; echo "bench dist rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 HIPK_DIST_OVERLAP=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist_ov.json 2> $O/bench_dist_ov.err; echo "bench dist overlap rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_dist","bench_dist_ov"):
    try:
        d=json.loads(open(f"gpurun_out/r02c19/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(1e6/d["value"]*1,2), "us/iter")
    except Exception as e: print(f, "ERR", e)
PY
