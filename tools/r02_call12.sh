#!/bin/bash
# round 2 evidence run: bench lines, rocprofv3 traces + PMC passes of the bench and of the GMRES / BiCGStab probes, harness report
set -o pipefail
O=gpurun_out/r02c12
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 500 bash tools/prof_bench.sh r02c12/prof_bench > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?" | tee -a $O/status.txt
timeout -k 10 500 bash tools/prof_solvers.sh r02c12/prof_solvers > $O/prof_solvers.log 2>&1; echo "prof_solvers rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python -m pytorch_sparse_solver.tests.benchmark --sparse --sizes 10000,1000000,4000000 --runs 2 --tol 1e-6 --maxiter 10000 --output-dir $O/report > $O/harness_sparse.log 2>&1; echo "harness sparse rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python -m pytorch_sparse_solver.tests.benchmark --quick --output-dir $O/report_quick > $O/harness_quick.log 2>&1; echo "harness quick rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/bench_reference_table.py > $O/reference_table.jsonl 2> $O/reference_table.err; echo "reftable rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python tools/bench_solvers.py 2000 2>/dev/null | grep "^{" > $O/solvers.jsonl; echo "solvers rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_bench $O/pmc_bench.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc bench json rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_solvers $O/pmc_solvers.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc solvers json rc=$?" | tee -a $O/status.txt
find $O -name "*counter_collection.csv" -delete
# keep what travels back small: drop the raw per-dispatch traces, keep stats + counter CSVs
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
