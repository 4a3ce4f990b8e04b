#!/bin/bash
# rocprofv3 of the GMRES(30) and BiCGStab probes at N = 4M: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in separate passes.
# usage (GPU box, repo root): bash tools/prof_solvers.sh <name>   -> gpurun_out/<name>/{gmres,bicgstab}/...
set -e
OUT=gpurun_out/${1:-prof_solvers}
export TMPDIR=/tmp
for P in gmres bicgstab; do
  mkdir -p $OUT/$P
  ARGS="tools/${P}_probe.py 2000"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$P/trace -- python3 $ARGS > $OUT/$P/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$P/pmc_fetch -- python3 $ARGS > $OUT/$P/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$P/pmc_write -- python3 $ARGS > $OUT/$P/pmc_write.log 2>&1
  python3 tools/summarize_prof.py $OUT/$P > $OUT/$P/summary.txt 2>&1 || true
  tail -1 $OUT/$P/trace.log
done
