#!/bin/bash
set -o pipefail
O=gpurun_out/r02c9
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "gmres" > $O/pytest.log 2>&1; echo "pytest gmres rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_dist.log
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep -v Warn | tee $O/small.log
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so HIPK_GM_STAMPS=1 timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep "stamps\|solve" | tee $O/small_stamps.log
HIPK_BENCH_DIST=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $O/bench_weak1_dist.json 2> $O/bench_weak1_dist.err; echo "bench dist rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 HIPK_DIST_NATIVE=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $O/bench_weak1_distpy.json 2> $O/bench_weak1_distpy.err; echo "bench dist py rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_weak1_dist","bench_weak1_distpy"):
    try:
        d=json.loads(open(f"gpurun_out/r02c9/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), d["ms_per_step"], d["config"]["iterations_per_solve"], 1e3*d["ms_per_step"]/d["config"]["iterations_per_solve"], "us/it")
    except Exception as e:
        print(f, "ERR", e); print(open(f"gpurun_out/r02c9/{f}.err").read()[-1500:])
PY
