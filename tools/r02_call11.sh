#!/bin/bash
set -o pipefail
O=gpurun_out/r02c11
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py tests/test_gpu_pcg.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_uni.json 2> $O/bench_uni.err; echo "bench uni rc=$?" | tee -a $O/status.txt
HIPK_SPMV_SELL_NO_UNI=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_nouni.json 2> $O/bench_nouni.err; echo "bench nouni rc=$?" | tee -a $O/status.txt
HIPK_SPMV_SELL_NO_PAIR=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_loop.json 2> $O/bench_loop.err; echo "bench loop rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_uni","bench_nouni","bench_loop"):
    d=json.loads(open(f"gpurun_out/r02c11/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), [ (k["key"], round(k["avg_launch_us"],2)) for k in d["kernels"]], round(d["spmv_standalone"]["us"],2))
PY
timeout -k 10 300 python tools/bench_solvers.py 2000 2>&1 | grep -v Warn > $O/solvers.jsonl; python - <<'PY'
import json
for ln in open("gpurun_out/r02c11/solvers.jsonl"):
    try:
        d=json.loads(ln); print(d["case"], "info", d["info"], "it", d["iterations_or_cycles"], "wall_ms", round(d["wall_ms"],2), "it/s", round(d["iters_per_s"],1))
    except Exception: pass
PY
