#!/bin/bash
# round 3, call 15: fused exchanges with the rank's own partials read in place: parity + world-1 rehearsal; pytree test
set -o pipefail
O=gpurun_out/r03c15
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py tests/test_gpu_api.py -m gpu -x -q -k "fused or pytree" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert\|rror:" $O/pytest.log | head -30; exit 1; }
for comm in rccl fused rccl fused; do
  for shape in "2000 0 2000" "1000 32768 8000 400"; do
    echo "== HIPK_DIST_COMM=$comm dist_probe $shape" | tee -a $O/probe.log
    HIPK_DIST_COMM=$comm timeout -k 10 300 python tools/dist_probe.py $shape 2>&1 | grep "us/iter\|single-device" | tee -a $O/probe.log
  done
done
