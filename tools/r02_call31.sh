#!/bin/bash
# round 2, call 31: row-partitioned solvers with rank-sized row blocks (two ranks on one GPU) on the two-rows-per-lane kernel, a rank
# of config 5 at 8 ranks (8 M rows, 8000-wide grid lines, chunk 32768) at world size 1, CG vector-kernel cache policy A/B at N = 32 M / 64 M
set -o pipefail
O=gpurun_out/r02c31
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q -k "large_row_blocks" > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest_dist.log | cut -c1-300
grep -q "pytest dist rc=0" $O/status.txt || exit 1
for st in 0 auto; do
  if [ $st = auto ]; then unset HIPK_SPMV_SELL_STRIDED; else export HIPK_SPMV_SELL_STRIDED=$st; fi
  echo "== config 5 rank of 8: 1000 x 8000 rows, chunk 32768, strided $st" >> $O/dist.log
  timeout -k 10 200 python tools/dist_probe.py 1000 32768 8000 2000 >> $O/dist.log 2>&1; echo "dist c5 $st rc=$?" | tee -a $O/status.txt
  echo "== config 5 rank of 4: 2000 x 8000 rows, chunk 32768, strided $st" >> $O/dist.log
  timeout -k 10 200 python tools/dist_probe.py 2000 32768 8000 2000 >> $O/dist.log 2>&1; echo "dist c5/4 $st rc=$?" | tee -a $O/status.txt
done
unset HIPK_SPMV_SELL_STRIDED
grep -E "==|dist_cg|chunk size" $O/dist.log
for cs in 0 1; do
  echo "== HIPK_CG_STREAMS=$cs" >> $O/streams.log
  HIPK_CG_STREAMS=$cs timeout -k 10 300 python tools/walk_probe.py 5657 8000 2>/dev/null | grep '"strided": null' >> $O/streams.log; echo "streams $cs rc=$?" | tee -a $O/status.txt
done
cut -c1-330 $O/streams.log
