#!/bin/bash
# round 3, call 13: grouped walk of the two-rows-per-lane SpMV with BOTH tiles of a wavefront pair in flight (HIPK_SPMV_SELL_PF=1,
# default) against one tile at a time (=0): parity, then N = 64 M / 32 M / 16 M and a rank-shaped block, separate processes, alternating
set -o pipefail
O=gpurun_out/r03c13
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_coded.py tests/test_distributed_gloo.py -m gpu -x -q -k "two_rows or many_grid_lines or large_row_blocks" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert" $O/pytest.log | head; exit 1; }
for rep in 1 2; do
  for pf in 1 0; do
    echo "== HIPK_SPMV_SELL_PF=$pf rep $rep" | tee -a $O/ab.log
    HIPK_SPMV_SELL_PF=$pf timeout -k 10 400 python tools/walk_probe.py 8000 5657 2>/dev/null | grep '"strided": null' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['nx'], d['kernel'][-10:], 'alone', d['spmv_alone_us'], 'incg', d['spmv_in_cg_us'], 'cg', d['cg_us_per_iter'], d['x_sha'])
" | tee -a $O/ab.log
    HIPK_SPMV_SELL_PF=$pf timeout -k 10 200 python tools/dist_probe.py 1000 32768 8000 400 2>/dev/null | grep "us/iter" | tee -a $O/ab.log
  done
done
