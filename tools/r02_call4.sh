#!/bin/bash
set -o pipefail
O=gpurun_out/r02c4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_pcg.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest.log
timeout -k 10 300 python tools/gmres_variants.py 2000 HIPK_GMRES_NO_STREAM=1,HIPK_GM_SPEC=0 HIPK_GM_SPEC=0,HIPK_GM_NRES=31 HIPK_GM_SPEC=0,HIPK_GM_NRES=0 HIPK_GM_SPEC=0,HIPK_GM_NRES=2 HIPK_GM_SPEC=0,HIPK_GM_NRES=4 HIPK_GM_SPEC=0,HIPK_GM_NRES=5 HIPK_GM_SPEC=0,HIPK_GM_NRES=6 \
   HIPK_GM_NRES=0 HIPK_GM_NRES=3 HIPK_GM_NRES=4 HIPK_GM_NRES=5 2>&1 | grep cycle | tee $O/variants.log
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | tee $O/small.log
