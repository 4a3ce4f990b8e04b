#!/bin/bash
set -o pipefail
O=gpurun_out/r02c3
mkdir -p $O
python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
python tools/gmres_variants.py 2000 HIPK_GMRES_NO_SWEEP=1 HIPK_GM_MAP=0,HIPK_GM_FOLD=0 HIPK_GM_MAP=0,HIPK_GM_FOLD=1 HIPK_GM_MAP=1,HIPK_GM_FOLD=0 HIPK_GM_MAP=1,HIPK_GM_FOLD=1 \
   HIPK_GM_FOLD=1,HIPK_GM_NRES=0 HIPK_GM_FOLD=1,HIPK_GM_NRES=3 HIPK_GM_FOLD=1,HIPK_GM_NRES=4 HIPK_GM_FOLD=1,HIPK_GM_NRES=5 HIPK_GM_FOLD=1,HIPK_GM_NRES=6 HIPK_GM_FOLD=1,HIPK_GM_NRES=7 \
   HIPK_GM_MAP=1,HIPK_GM_FOLD=1,HIPK_GM_NRES=5 HIPK_GM_FOLD=0,HIPK_GM_NRES=5 2>&1 | tee $O/variants.log
