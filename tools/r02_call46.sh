#!/bin/bash
# round 2, call 46: tiles per workgroup of the grouped walk: 4 / 8 / 16 (library twins built with -DHIPK_SELL_GROUP), same box, alternating
set -o pipefail
O=gpurun_out/r02c47
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
L=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib
for rep in 1 2; do
  for g in 4 2 8; do
    if [ $g = 8 ]; then unset HIPK_LIB_PATH; else export HIPK_LIB_PATH=$L/libhipk_g$g.so; fi
    echo "== group $g rep $rep" | tee -a $O/groups.log
    timeout -k 10 300 python tools/walk_probe.py 5657 8000 2>/dev/null | grep '"strided": null' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['nx'], d['kernel'][-10:], 'alone', d['spmv_alone_us'], 'incg', d['spmv_in_cg_us'], 'cg', d['cg_us_per_iter'], d['x_sha'])
" | tee -a $O/groups.log
    timeout -k 10 200 python tools/dist_probe.py 2000 16384 2>/dev/null | grep dist_cg | tee -a $O/groups.log
  done
done
