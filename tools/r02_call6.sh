#!/bin/bash
set -o pipefail
O=gpurun_out/r02c6
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
HIPK_GM_STAMPS=1 timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep -v Warn | tee $O/small.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_pair.json 2> $O/bench_pair.err; echo "bench pair rc=$?" | tee -a $O/status.txt
HIPK_SPMV_SELL_NO_PAIR=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_nopair.json 2> $O/bench_nopair.err; echo "bench nopair rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_pair","bench_nopair"):
    d=json.loads(open(f"gpurun_out/r02c6/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), [ (k["key"], round(k["avg_launch_us"],2)) for k in d["kernels"]], round(d["spmv_standalone"]["us"],2))
PY
