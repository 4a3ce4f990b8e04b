#!/bin/bash
# round 2, call 34: BASELINE config 5's system on one device against the reference's fixture (N = 64 M)
set -o pipefail
O=gpurun_out/r02c34
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "config5 or full_size_against" > $O/pytest_c5.log 2>&1; echo "pytest c5 rc=$?" | tee -a $O/status.txt
grep -E "poisson_nx|convdiff_nx|passed|failed|Error|assert" $O/pytest_c5.log | cut -c1-300
