#!/bin/bash
# round 3, call 7: coded SpMV (two rows per lane) with the uniform tiles BEFORE the dictionary barrier (scalar dictionary loads)
# against the committed kernel (library twin libhipk_head.so), same box, alternating; parity of the new one first
set -o pipefail
O=gpurun_out/r03c12
mkdir -p $O
export TMPDIR=/tmp
L=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib
timeout -k 10 900 python -m pytest tests/test_gpu_coded.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || exit 1
for rep in 1 2 3; do
  for v in head new; do
    if [ $v = new ]; then unset HIPK_LIB_PATH; else export HIPK_LIB_PATH=$L/libhipk_head.so; fi
    for nx in 2000 1400; do
      echo -n "$v rep $rep: " | tee -a $O/ab.log
      timeout -k 10 200 python tools/spmv_wide_probe.py $nx 2>/dev/null | tail -1 | tee -a $O/ab.log
    done
  done
done
