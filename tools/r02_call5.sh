#!/bin/bash
set -o pipefail
O=gpurun_out/r02c5
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python tools/gmres_variants.py 2000 HIPK_GMRES_NO_STREAM=1,HIPK_GM_SPEC=0 HIPK_GM_SPEC=0,HIPK_GM_NRES=31 HIPK_GM_SPEC=0,HIPK_GM_NRES=5 HIPK_GM_SPEC=0,HIPK_GM_NRES=31,HIPK_GM_SWEEP3=1 HIPK_GM_SPEC=0,HIPK_GM_NRES=5,HIPK_GM_SWEEP3=1 HIPK_GM_SPEC=0,HIPK_GM_NRES=0,HIPK_GM_SWEEP3=1 HIPK_GM_NRES=5,HIPK_GM_SWEEP3=1 2>&1 | grep cycle | tee $O/variants.log
HIPK_GM_SPEC=0 HIPK_GM_NRES=5 HIPK_GM_SWEEP3=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_sweep3 -- python3 tools/gmres_probe.py 2000 batched > $O/tr_sweep3.log 2>&1
HIPK_GM_SPEC=0 HIPK_GM_NRES=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_stream -- python3 tools/gmres_probe.py 2000 batched > $O/tr_stream.log 2>&1
python3 tools/summarize_prof.py $O/tr_sweep3 | head -8
python3 tools/summarize_prof.py $O/tr_stream | head -8
