#!/bin/bash
mkdir -p gpurun_out/r03c24
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "mid_one_launch or whole_loop_in_one_launch or two_launch" > gpurun_out/r03c24/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r03c24/pytest.log
[ $rc -eq 0 ] || exit 1
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/cg_mid_stamps_probe.py > gpurun_out/r03c24/stamps.jsonl 2> gpurun_out/r03c24/stamps.err
echo "stamps rc=$?"; cat gpurun_out/r03c24/stamps.jsonl
timeout -k 10 300 python tools/cg_mid_probe.py > gpurun_out/r03c24/cg_mid.jsonl 2> gpurun_out/r03c24/cg_mid.err
echo "probe rc=$?"; grep mid gpurun_out/r03c24/cg_mid.jsonl
