#!/bin/bash
set -o pipefail
O=gpurun_out/r02c10
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest.log
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep -v Warn | tee $O/small.log
for S in 1 0; do
HIPK_CG_STREAMS=$S timeout -k 10 300 python bench.py --scaling strong --steps 1 --warmup 1 > $O/bench_strong_streams$S.json 2> $O/bench_strong_streams$S.err; echo "strong streams=$S rc=$?" | tee -a $O/status.txt
done
python - <<'PY'
import json
for f in ("bench_strong_streams1","bench_strong_streams0"):
    try:
        d=json.loads(open(f"gpurun_out/r02c10/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],1), [ (k["key"], round(k["avg_launch_us"],1)) for k in d["kernels"]])
    except Exception as e:
        print(f, "ERR", e)
PY
