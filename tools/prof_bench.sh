#!/bin/bash
# rocprofv3 of the bench command: kernel-trace stats, then PMC counters in separate passes (no trace domains mixed in).
# usage (GPU box, repo root): bash tools/prof_bench.sh <name>   -> gpurun_out/<name>/
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/pmc_l2.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
tail -3 $OUT/trace.log
