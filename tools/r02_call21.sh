#!/bin/bash
# round 2, evidence refresh after the small-system CG / BiCGStab kernels: full GPU suite, solver tables, small-system probes
# (+ rocprofv3 kernel traces of them), harness reports, reference table
set -o pipefail
O=gpurun_out/r02c21
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_gpu.log
grep -q "pytest gpu rc=0" $O/status.txt || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python tools/bench_solvers.py 2000 2>/dev/null | grep "^{" > $O/solvers.jsonl; echo "solvers rc=$?" | tee -a $O/status.txt
timeout -k 10 100 python tools/small_cg_probe.py 2>&1 | grep "solve" | tee $O/small_cg.log
timeout -k 10 100 python tools/small_gmres_probe.py 2>&1 | grep "solve" | tee $O/small_gmres.log
timeout -k 10 100 python tools/overhead_probe.py 2>&1 | grep "n=" | tee $O/overhead.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small_cg -- python3 tools/small_cg_probe.py > $O/prof_small_cg.log 2>&1; echo "prof small cg rc=$?" | tee -a $O/status.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small_gmres -- python3 tools/small_gmres_probe.py > $O/prof_small_gmres.log 2>&1; echo "prof small gmres rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/bench_reference_table.py > $O/reference_table.jsonl 2> $O/reference_table.err; echo "reftable rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python -m pytorch_sparse_solver.tests.benchmark --sparse --sizes 10000,1000000,4000000 --runs 2 --tol 1e-6 --maxiter 10000 --output-dir $O/report > $O/harness_sparse.log 2>&1; echo "harness sparse rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python -m pytorch_sparse_solver.tests.benchmark --quick --output-dir $O/report_quick > $O/harness_quick.log 2>&1; echo "harness quick rc=$?" | tee -a $O/status.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in prof_small_cg prof_small_gmres; do f=$(ls $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== $d"; head -8 "$f" | cut -c1-200; done
du -sh $O
