#!/bin/bash
# rocprofv3 kernel trace of the bench command only (no PMC passes). usage: bash tools/prof_trace_only.sh <name>
set -e
OUT=gpurun_out/${1:-trace}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
tail -2 $OUT/trace.log
head -12 $OUT/summary.txt
