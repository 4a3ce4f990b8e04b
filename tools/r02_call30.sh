#!/bin/bash
# round 2, call 30: final evidence on the final code: rocprofv3 trace + PMC passes of the bench -> per-kernel traffic json (taken
# BEFORE the bench line, so that the line's `traffic` refers to the same code), bench line, solver tables + solver profiles
set -o pipefail
O=gpurun_out/r02c30
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 400 python -m pytest tests/test_gpu_coded.py -m gpu -x -q > $O/pytest_coded.log 2>&1; echo "pytest coded rc=$?" | tee -a $O/status.txt
tail -1 $O/pytest_coded.log
grep -q "pytest coded rc=0" $O/status.txt || exit 1
timeout -k 10 500 bash tools/prof_bench.sh r02c30/prof_bench > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_bench $O/pmc_bench.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc bench json rc=$?" | tee -a $O/status.txt
cp $O/pmc_bench.json profiles/r02_pmc_kernels.json
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python tools/bench_solvers.py 2000 2>/dev/null | grep "^{" > $O/solvers.jsonl; echo "solvers rc=$?" | tee -a $O/status.txt
timeout -k 10 400 bash tools/prof_solvers.sh r02c30/prof_solvers > $O/prof_solvers.log 2>&1; echo "prof_solvers rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_solvers $O/pmc_solvers.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc solvers json rc=$?" | tee -a $O/status.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02c30/bench_line.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "it/s; roofline frac", round(d["roofline"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "cpu", round(d["cpu_baseline"]["value"]), d["kernels"][0]["kernel"][:50], d["kernels"][0]["traffic"])
PY
du -sh $O
