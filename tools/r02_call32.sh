#!/bin/bash
# round 2, call 32: the grouped walk for the one-row-per-lane coded kernels (fp32 storage, offset-coded form): tests, A/B per size
set -o pipefail
O=gpurun_out/r02c32
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 600 python -m pytest tests/test_gpu_coded.py -m gpu -x -q > $O/pytest_coded.log 2>&1; echo "pytest coded rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_coded.log | cut -c1-300
grep -q "pytest coded rc=0" $O/status.txt || exit 1
timeout -k 10 600 python tools/walk_probe.py f32:4000 f32:5657 f32:8000 var:2828 var:4000 var:5657 var:8000 > $O/walk.log 2>&1; echo "walk rc=$?" | tee -a $O/status.txt
grep -E '^\{|^nx ' $O/walk.log | cut -c1-420
