#!/bin/bash
# round 2, call 50: grouped walk with two tiles in flight per wavefront pair (92 VGPRs) against the committed one (37 VGPRs), library
# twins, same box, alternating; parity of the new one first
set -o pipefail
O=gpurun_out/r02c50
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
L=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib
timeout -k 10 600 python -m pytest tests/test_gpu_coded.py -m gpu -x -q -k "two_rows or many_grid_lines" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || exit 1
for rep in 1 2; do
  for v in head new; do
    if [ $v = new ]; then unset HIPK_LIB_PATH; else export HIPK_LIB_PATH=$L/libhipk_head.so; fi
    echo "== $v rep $rep" | tee -a $O/ab.log
    timeout -k 10 300 python tools/walk_probe.py 5657 8000 2>/dev/null | grep '"strided": null' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['nx'], d['kernel'][-10:], 'alone', d['spmv_alone_us'], 'incg', d['spmv_in_cg_us'], 'cg', d['cg_us_per_iter'], d['x_sha'])
" | tee -a $O/ab.log
    timeout -k 10 200 python tools/dist_probe.py 2000 16384 2>/dev/null | grep dist_cg | tee -a $O/ab.log
  done
done
