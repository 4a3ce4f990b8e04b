#!/bin/bash
# round 2, final evidence (part 1): full GPU test suite, smoke, bench line, rocprofv3 trace + PMC passes of the bench
set -o pipefail
O=gpurun_out/r02c17
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_gpu.log
grep -q "pytest gpu rc=0" $O/status.txt || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 400 bash tools/prof_bench.sh r02c17/prof_bench > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_bench $O/pmc_bench.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc bench json rc=$?" | tee -a $O/status.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02c17/bench_line.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "it/s; roofline frac", round(d["roofline"]["frac"],3), "traffic", d["roofline"].get("traffic"), "cpu", round(d["cpu_baseline"]["value"]))
PY
du -sh $O
