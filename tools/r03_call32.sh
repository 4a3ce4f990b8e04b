#!/bin/bash
mkdir -p gpurun_out/r03c32
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "cg_mid or cg_whole_loop or fall_back" > gpurun_out/r03c32/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v "Warning\|warn\|return torch\|^$\|Docs\|mid case" gpurun_out/r03c32/pytest.log | tail -8 | cut -c1-600
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/cg_mid_probe.py mid > gpurun_out/r03c32/cg_mid.jsonl 2> gpurun_out/r03c32/cg_mid.err
echo "cg probe rc=$?"; awk 'NR%2==0' gpurun_out/r03c32/cg_mid.jsonl
timeout -k 10 300 python tools/cg_mid3d_probe.py > gpurun_out/r03c32/cg_mid3d.jsonl 2> gpurun_out/r03c32/cg_mid3d.err
echo "3d probe rc=$?"; awk 'NR%4==3 || NR%4==0' gpurun_out/r03c32/cg_mid3d.jsonl; tail -3 gpurun_out/r03c32/cg_mid3d.err
