#!/bin/bash
# rocprofv3 passes for the SpMV kernel: kernel-trace stats, then PMC counters in separate runs.
# usage (on the GPU box, from repo root): bash tools/prof_spmv.sh <outdir-under-gpurun_out>
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 tools/gpu_probe.py 2000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
