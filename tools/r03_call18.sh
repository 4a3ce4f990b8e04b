#!/bin/bash
# round 3, call 18: GMRES(30) normalise step as one kernel / two launches over the system size (where does the split start to pay?)
set -o pipefail
O=gpurun_out/r03c18
mkdir -p $O
export TMPDIR=/tmp
for nx in 300 500 720 1000 1400; do
  timeout -k 10 200 python tools/gmres_probe3.py $nx HIPK_GM_SPLIT_NORM 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['nx'], d['nx'] ** 2 // 2048 + 1, 'split', d['value'], round(d['ms_per_cycle'], 4), d['x_sha'])" | tee -a $O/split.log
done
