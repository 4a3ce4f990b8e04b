#!/bin/bash
# round 2, call 40: same-box A/B of the vector kernels' batched streaming tail: the previous commit's library (libhipk_prev.so, built
# from `git archive HEAD~1`) against the current one, alternating, config 5 on one device
set -o pipefail
O=gpurun_out/r02c40
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
PREV=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_prev.so
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export HIPK_LIB_PATH=$PREV; else unset HIPK_LIB_PATH; fi
    timeout -k 10 200 python bench.py --scaling strong --steps 1 --warmup 0 --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; echo "$v $rep rc=$?" >> $O/status.txt
    python - $O/b_${v}_$rep.json $v $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], round(d["value"], 1), "it/s", [(k["key"], round(k["avg_launch_us"], 1)) for k in d["kernels"]], flush=True)
PY
  done
done
unset HIPK_LIB_PATH
