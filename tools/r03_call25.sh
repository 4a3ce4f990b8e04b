#!/bin/bash
mkdir -p gpurun_out/r03c25
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "mid_one_launch" > gpurun_out/r03c25/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r03c25/pytest.log
[ $rc -eq 0 ] || exit 1
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/cg_mid_stamps_probe.py > gpurun_out/r03c25/stamps.jsonl 2> gpurun_out/r03c25/stamps.err
echo "stamps rc=$?"; cat gpurun_out/r03c25/stamps.jsonl
timeout -k 10 300 python tools/cg_mid_probe.py mid,mid-s1,mid-s4,mid-s8 > gpurun_out/r03c25/cg_mid.jsonl 2> gpurun_out/r03c25/cg_mid.err
echo "probe rc=$?"; cat gpurun_out/r03c25/cg_mid.jsonl
