#!/bin/bash
# A/B of staggered polls (library twin built with -DHIPK_LL_STAGGER) on the three one-launch loops, same box, alternating
mkdir -p gpurun_out/r03c34
T=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stagger.so
for rep in 1 2; do
  for lib in default stagger; do
    if [ $lib = stagger ]; then export HIPK_LIB_PATH=$T; else unset HIPK_LIB_PATH; fi
    timeout -k 10 200 python tools/cg_mid_probe.py mid 2>/dev/null | awk 'NR%2==0' | sed "s/^/$lib cg /" >> gpurun_out/r03c34/ab.txt
    timeout -k 10 200 python tools/bicgstab_mid_probe.py 2>/dev/null | grep '"mid"' | awk 'NR%2==0' | sed "s/^/$lib bicgstab /" >> gpurun_out/r03c34/ab.txt
    timeout -k 10 200 python tools/gmres_mid_probe.py 300 500 720 2>/dev/null | grep '"mid"' | awk 'NR%2==0' | sed "s/^/$lib gmres /" >> gpurun_out/r03c34/ab.txt
  done
done
cat gpurun_out/r03c34/ab.txt | cut -c1-150
