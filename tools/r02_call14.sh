#!/bin/bash
set -o pipefail
O=gpurun_out/r02c14
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q -k "mailboxes" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
HIPK_BENCH_DIST=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist_rccl.json 2> $O/bench_dist_rccl.err; echo "bench rccl rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 HIPK_DIST_COMM=p2p timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist_p2p.json 2> $O/bench_dist_p2p.err; echo "bench p2p rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_dist_rccl","bench_dist_p2p"):
    try:
        d=json.loads(open(f"gpurun_out/r02c14/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), d["config"].get("collectives"), d["config"].get("rccl_ranks"))
    except Exception as e: print(f, "ERR", e)
PY
