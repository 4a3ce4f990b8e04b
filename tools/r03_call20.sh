#!/bin/bash
mkdir -p gpurun_out/r03c20
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so timeout -k 10 300 python tools/cg_mid_stamps_probe.py > gpurun_out/r03c20/stamps.jsonl 2> gpurun_out/r03c20/stamps.err
echo "rc=$?"; cat gpurun_out/r03c20/stamps.jsonl; tail -5 gpurun_out/r03c20/stamps.err
