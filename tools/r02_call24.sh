#!/bin/bash
# round 2, call 24: strided walk of the two-rows-per-lane coded SpMV (chunks of several grid lines: N = 16 M / 64 M), A/B per size;
# then the whole GPU suite on this code
set -o pipefail
O=gpurun_out/r02c24
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 400 python -m pytest tests/test_gpu_coded.py -m gpu -x -q -k "two_rows or many_grid_lines" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_new.log
grep -q "pytest new rc=0" $O/status.txt || exit 1
timeout -k 10 400 python tools/walk_probe.py 2000 2828 4000 5657 8000 > $O/walk.log 2>&1; echo "walk rc=$?" | tee -a $O/status.txt
cat $O/walk.log | cut -c1-330
grep -q "walk rc=0" $O/status.txt || exit 1
HIPK_SPMV_SELL_CHUNKED=0 timeout -k 10 300 python tools/walk_probe.py 4000 8000 > $O/walk_unchunked.log 2>&1; echo "walk unchunked rc=$?" | tee -a $O/status.txt
grep '"strided": null' $O/walk_unchunked.log | cut -c1-330
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_all.log
