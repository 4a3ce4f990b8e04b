#!/bin/bash
mkdir -p gpurun_out/r03c33
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -s -m gpu -k "gmres_mid" > gpurun_out/r03c33/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v "Warning\|warn\|return torch\|^$\|Docs" gpurun_out/r03c33/pytest.log | tail -25 | cut -c1-600
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gmres_mid_probe.py > gpurun_out/r03c33/gm_mid.jsonl 2> gpurun_out/r03c33/gm_mid.err
echo "probe rc=$?"; cat gpurun_out/r03c33/gm_mid.jsonl
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/gmres_mid_stamps_probe.py 500 > gpurun_out/r03c33/stamps.jsonl 2> gpurun_out/r03c33/stamps.err
echo "stamps rc=$?"; cat gpurun_out/r03c33/stamps.jsonl
