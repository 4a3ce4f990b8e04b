#!/bin/bash
# round 2, GPU call 1: full GPU suite, default bench line, config-5 N=1 legs (single-device cg and the row-partitioned code at world 1)
set -o pipefail
O=gpurun_out/r02c1
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/status.txt
python bench.py --scaling strong --steps 2 --warmup 1 > $O/bench_strong1.json 2> $O/bench_strong1.err; echo "bench strong rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 python bench.py --scaling strong --steps 2 --warmup 1 > $O/bench_strong1_dist.json 2> $O/bench_strong1_dist.err; echo "bench strong dist rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 python bench.py --steps 3 --warmup 1 > $O/bench_weak1_dist.json 2> $O/bench_weak1_dist.err; echo "bench weak dist rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
