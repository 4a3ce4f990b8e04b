#!/bin/bash
# round 3, call 6: exchanges fused into the CG kernels -- parity with ranks sharing cuda:0, then the world-1 rehearsal of the
# loop for a 4 M-row block and for config 5's rank block (1000 grid lines of 8000, chunk 32768): rccl / p2p / fused / single device
set -o pipefail
O=gpurun_out/r03c6
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py tests/test_distributed_api.py tests/test_abi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert\|rror:" $O/pytest.log | head -30; exit 1; }
for comm in rccl p2p fused; do
  for shape in "2000 0 2000" "1000 32768 8000 400"; do
    echo "== HIPK_DIST_COMM=$comm dist_probe $shape" | tee -a $O/probe.log
    HIPK_DIST_COMM=$comm timeout -k 10 300 python tools/dist_probe.py $shape 2>&1 | grep -v "^\[\|NCCL\|rccl" | tee -a $O/probe.log
  done
done
