#!/bin/bash
# round 2, call 25: the free-grid strided walk of the two-rows-per-lane coded SpMV: parity tests, A/B per size (with / without
# non-temporal y), a rank's row block of a larger system (forced chunk sizes of 4- and 8-rank weak scaling) at world size 1
set -o pipefail
O=gpurun_out/r02c25
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 400 python -m pytest tests/test_gpu_coded.py -m gpu -x -q -k "two_rows or many_grid_lines" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_new.log
grep -q "pytest new rc=0" $O/status.txt || exit 1
timeout -k 10 500 python tools/walk_probe.py 700 1000 1400 2000 4000 8000 > $O/walk.log 2>&1; echo "walk rc=$?" | tee -a $O/status.txt
grep -v Warning $O/walk.log | cut -c1-360
grep -q "walk rc=0" $O/status.txt || exit 1
for ch in 8192 16384; do
  for st in 0 auto; do
    if [ $st = auto ]; then unset HIPK_SPMV_SELL_STRIDED; else export HIPK_SPMV_SELL_STRIDED=$st; fi
    echo "== force_ch $ch strided $st" >> $O/dist.log
    timeout -k 10 200 python tools/dist_probe.py 2000 $ch >> $O/dist.log 2>&1; echo "dist $ch $st rc=$?" | tee -a $O/status.txt
  done
done
unset HIPK_SPMV_SELL_STRIDED
grep -E "==|dist_cg|chunk size" $O/dist.log
