#!/bin/bash
# round 3, call 16: two launches per CG iteration on mid-size systems: parity, then per-iteration times against three launches
set -o pipefail
O=gpurun_out/r03c17
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q -k "two_launch or cg" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert\|rror:" $O/pytest.log | head -30; exit 1; }
timeout -k 10 400 python tools/cg_mid_probe.py > $O/cg_mid.jsonl 2> $O/cg_mid.err; echo "probe rc=$?" | tee -a $O/status.txt
python3 - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/r03c17/cg_mid.jsonl"):
    d = json.loads(l); acc[(d["n"], d["chunks"], d["two_launch"])].append(d["us_per_iteration"])
for k in sorted(acc): print(k, acc[k])
PY
