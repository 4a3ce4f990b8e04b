#!/bin/bash
# call 19: one-launch CG loop for mid-size systems -- parity test, then per-iteration times
mkdir -p gpurun_out/r03c19
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "mid_one_launch or two_launch or small" > gpurun_out/r03c19/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r03c19/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/cg_mid_probe.py > gpurun_out/r03c19/cg_mid.jsonl 2> gpurun_out/r03c19/cg_mid.err
echo "probe rc=$?"; cat gpurun_out/r03c19/cg_mid.jsonl
