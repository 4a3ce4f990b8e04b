#!/bin/bash
mkdir -p gpurun_out/r03c28
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "cg_mid or cg_whole_loop or fall_back or bicgstab_mid" > gpurun_out/r03c28/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v "Warning\|warn\|return torch\|^$\|Docs\|bicgstab mid case" gpurun_out/r03c28/pytest.log | tail -12 | cut -c1-600
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/cg_mid_probe.py mid,0 > gpurun_out/r03c28/cg_mid.jsonl 2> gpurun_out/r03c28/cg_mid.err
echo "probe rc=$?"; grep '"mid"' gpurun_out/r03c28/cg_mid.jsonl
timeout -k 10 300 python tools/bicgstab_mid_probe.py > gpurun_out/r03c28/bi_mid.jsonl 2> gpurun_out/r03c28/bi_mid.err
echo "probe rc=$?"; grep '"mid"' gpurun_out/r03c28/bi_mid.jsonl
