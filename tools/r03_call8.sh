#!/bin/bash
# round 3, call 8: per-phase stamps of the two-rows-per-lane coded SpMV (diagnostic twin)
set -o pipefail
O=gpurun_out/r03c10
mkdir -p $O
export TMPDIR=/tmp
L=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib
HIPK_LIB_PATH=$L/libhipk_stamps.so timeout -k 10 300 python tools/spmv_stamps_probe.py 2000 > $O/stamps_nx2000.jsonl 2> $O/stamps.err; echo "stamps rc=$?" | tee -a $O/status.txt
cat $O/stamps_nx2000.jsonl | cut -c1-900
