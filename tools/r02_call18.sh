#!/bin/bash
# round 2, final evidence (part 2): solver tables, small-system GMRES timings + stamps, solver profiles, harness reports
set -o pipefail
O=gpurun_out/r02c18
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 300 python tools/bench_solvers.py 2000 2>/dev/null | grep "^{" > $O/solvers.jsonl; echo "solvers rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/small_gmres_probe.py 2>&1 | grep "solve" | tee $O/small.log
HIPK_LIB_PATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk_stamps.so HIPK_GM_STAMPS=1 timeout -k 10 200 python tools/small_gmres_probe.py > $O/stamps.log 2>&1; echo "stamps rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/gmres_variants.py 2000 HIPK_GMRES_NO_STREAM=1,HIPK_GM_SPEC=0 "" 2>&1 | grep cycle | tee $O/variants.log
timeout -k 10 200 python tools/bench_reference_table.py > $O/reference_table.jsonl 2> $O/reference_table.err; echo "reftable rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python -m pytorch_sparse_solver.tests.benchmark --sparse --sizes 10000,1000000,4000000 --runs 2 --tol 1e-6 --maxiter 10000 --output-dir $O/report > $O/harness_sparse.log 2>&1; echo "harness sparse rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python -m pytorch_sparse_solver.tests.benchmark --quick --output-dir $O/report_quick > $O/harness_quick.log 2>&1; echo "harness quick rc=$?" | tee -a $O/status.txt
timeout -k 10 400 bash tools/prof_solvers.sh r02c18/prof_solvers > $O/prof_solvers.log 2>&1; echo "prof_solvers rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_solvers $O/pmc_solvers.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc solvers json rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist_rccl.json 2> $O/bench_dist_rccl.err; echo "bench dist rccl rc=$?" | tee -a $O/status.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
