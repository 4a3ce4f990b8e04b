#!/bin/bash
mkdir -p gpurun_out/r03c31
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "mid" > gpurun_out/r03c31/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v "Warning\|warn\|return torch\|^$\|Docs\|mid case" gpurun_out/r03c31/pytest.log | tail -8 | cut -c1-600
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/cg_mid_probe.py mid > gpurun_out/r03c31/cg_mid.jsonl 2> gpurun_out/r03c31/cg_mid.err
echo "cg probe rc=$?"; awk 'NR%2==0' gpurun_out/r03c31/cg_mid.jsonl
timeout -k 10 300 python tools/bicgstab_mid_probe.py > gpurun_out/r03c31/bi_mid.jsonl 2> gpurun_out/r03c31/bi_mid.err
echo "bicgstab probe rc=$?"; grep '"mid"' gpurun_out/r03c31/bi_mid.jsonl | awk 'NR%2==0'
timeout -k 10 300 python tools/gmres_mid_probe.py > gpurun_out/r03c31/gm_mid.jsonl 2> gpurun_out/r03c31/gm_mid.err
echo "gmres probe rc=$?"; awk 'NR%4==3 || NR%4==0' gpurun_out/r03c31/gm_mid.jsonl
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/gmres_mid_stamps_probe.py 500 > gpurun_out/r03c31/gm_stamps.jsonl 2> gpurun_out/r03c31/gm_stamps.err
echo "stamps rc=$?"; cat gpurun_out/r03c31/gm_stamps.jsonl
