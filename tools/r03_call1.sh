#!/bin/bash
# round 3, call 1: GPU suite on the rebuilt library (kernel durations now from start/stop events bound to the dispatch), bench.py,
# and rocprofv3 --kernel-trace --stats of the same bench command to check that its averages equal the in-loop figures
set -o pipefail
O=gpurun_out/r03c4
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
grep -q "bench rc=0" $O/status.txt || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c4/bench.json"))
print("iter us", d["roofline"]["iteration_us_sum_of_kernels"], d["roofline"]["iteration_us_from_timed_region"]); print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["config"].get("cold_first_solve_ms"))
for k in d["kernels"]:
    print(k["key"], round(k["avg_launch_us"], 2), "us chain,", round(k["dispatch_span_us"], 2), "span", round(k["frac_of_hbm_peak"], 3))
for name, leg in d["roofline"]["legs"].items():
    if isinstance(leg, dict):
        print(name, round(leg["avg_launch_us"], 2), "us frac", round(leg["frac"], 3), leg["kernel"][:60])
    else:
        for k in leg:
            print("  n64m", k["key"], round(k["avg_launch_us"], 2), "us", round(k["frac_of_hbm_peak"], 3), k.get("scalars_launch_us"))
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "trace rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/status.txt
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py $O/trace > $O/trace_summary.txt 2>&1 || true
head -16 $O/trace_summary.txt | cut -c1-220
tail -1 $O/trace.log | cut -c1-300
timeout -k 10 400 python tools/gmres_probe3.py 2000 > $O/gmres_probe3.jsonl 2> $O/gmres_probe3.err; echo "gmres probe rc=$?" | tee -a $O/status.txt
cat $O/gmres_probe3.jsonl | cut -c1-200
