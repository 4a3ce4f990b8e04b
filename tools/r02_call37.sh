#!/bin/bash
# round 2, call 36: batched streaming tail of the vector kernels (hipk_pre): parity, then N = 32 M / 64 M timings
set -o pipefail
O=gpurun_out/r02c37
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "multi_step_chunks or config5 or n4m_headline" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
grep -E "poisson_nx|passed|failed|Error|assert" $O/pytest.log | cut -c1-300
grep -q "pytest rc=0" $O/status.txt || exit 1
timeout -k 10 300 python tools/walk_probe.py 4000 5657 8000 2>/dev/null | grep '"strided": null' > $O/walk.log; echo "walk rc=$?" | tee -a $O/status.txt
cut -c1-330 $O/walk.log
timeout -k 10 300 python bench.py --scaling strong --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_strong_n1.json 2> $O/bench_strong_n1.err; echo "bench strong rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02c37/bench_strong_n1.json").read().strip().splitlines()[-1])
print("strong n1", round(d["value"], 1), "it/s", [(k["key"], round(k["avg_launch_us"], 1), round(k.get("frac_of_hbm_peak") or 0, 3)) for k in d["kernels"]])
PY
