#!/bin/bash
# round 3, call 11: SGPR caps (8 workgroups per CU where the hardware admitted 7): parity of the touched kernels, GMRES cycle A/B
# against the previous library (twin libhipk_head.so = round-3 head before the caps), fused-exchange rehearsal, stamps with the fixed twin
set -o pipefail
O=gpurun_out/r03c11
mkdir -p $O
export TMPDIR=/tmp
L=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib
timeout -k 10 900 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py tests/test_distributed_gloo.py -m gpu -x -q -k "gmres or fused or two_rows or coded or stencil or wide" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert\|rror:" $O/pytest.log | head -30; exit 1; }
for rep in 1 2; do
  for v in head new; do
    if [ $v = new ]; then unset HIPK_LIB_PATH; else export HIPK_LIB_PATH=$L/libhipk_head.so; fi
    echo "== $v rep $rep" | tee -a $O/gmres_ab.log
    timeout -k 10 300 python tools/gmres_probe3.py 2000 HIPK_GM_DUMMY 2>/dev/null | grep '"value": "1"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['restart'], round(d['ms_per_cycle'], 3), d['x_sha'])" | tee -a $O/gmres_ab.log
  done
done
unset HIPK_LIB_PATH
for comm in rccl fused; do
  for shape in "2000 0 2000" "1000 32768 8000 400"; do
    echo "== HIPK_DIST_COMM=$comm dist_probe $shape" | tee -a $O/probe.log
    HIPK_DIST_COMM=$comm timeout -k 10 300 python tools/dist_probe.py $shape 2>&1 | grep "us/iter\|single-device" | tee -a $O/probe.log
  done
done
HIPK_LIB_PATH=$L/libhipk_stamps.so timeout -k 10 300 python tools/spmv_stamps_probe.py 2000 > $O/stamps_nx2000.jsonl 2> $O/stamps.err; echo "stamps rc=$?" | tee -a $O/status.txt
tail -2 $O/stamps_nx2000.jsonl | cut -c1-1400
