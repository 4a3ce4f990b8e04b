#!/bin/bash
mkdir -p gpurun_out/r03c27
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -s -k "bicgstab_mid or bicgstab_whole_loop or fall_back" > gpurun_out/r03c27/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r03c27/pytest.log | cut -c1-400
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bicgstab_mid_probe.py > gpurun_out/r03c27/bi_mid.jsonl 2> gpurun_out/r03c27/bi_mid.err
echo "probe rc=$?"; cat gpurun_out/r03c27/bi_mid.jsonl
