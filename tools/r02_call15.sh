#!/bin/bash
set -o pipefail
O=gpurun_out/r02c15
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "gmres" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | tee $O/small.log
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so HIPK_GM_STAMPS=1 timeout -k 10 200 python tools/small_gmres_probe.py > $O/stamps.log 2>&1; echo "stamps rc=$?" | tee -a $O/status.txt
grep -h "stamps\|abandoned" $O/stamps.log | head -8
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_parity.log
