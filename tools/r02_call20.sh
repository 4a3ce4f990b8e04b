#!/bin/bash
set -o pipefail
O=gpurun_out/r02c20
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "whole_loop" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -8 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
timeout -k 10 100 python tools/small_cg_probe.py 2>&1 | grep "solve" | tee $O/small_cg.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pcg.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_parity.log
