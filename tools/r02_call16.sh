#!/bin/bash
set -o pipefail
O=gpurun_out/r02c16
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep "solve" | tee $O/small.log
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so HIPK_GM_STAMPS=1 timeout -k 10 200 python tools/small_gmres_probe.py > $O/stamps.log 2>&1; echo "stamps rc=$?" | tee -a $O/status.txt
grep -h -A1 "stamps" $O/stamps.log | head -4
