#!/bin/bash
mkdir -p gpurun_out/r03c30
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/gmres_mid_stamps_probe.py > gpurun_out/r03c30/stamps.jsonl 2> gpurun_out/r03c30/stamps.err
echo "stamps rc=$?"; cat gpurun_out/r03c30/stamps.jsonl; tail -3 gpurun_out/r03c30/stamps.err
