#!/bin/bash
# round 3, call 5: full GPU suite (matrix-free loops, restart > 31, row-block operand, split normalise, placement probe), bench,
# rocprofv3 kernel trace of the bench command -> profile digest, GMRES A/B of the split normalise step, config 5 on one device x 3
set -o pipefail
O=gpurun_out/r03c5
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert" $O/pytest.log | head -20; exit 1; }
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
grep -q "bench rc=0" $O/status.txt || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c5/bench.json"))
r = d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["config"].get("cold_first_solve_ms"))
print("iter us", r["iteration_us_from_timed_region"], "sum events", r["iteration_us_sum_of_event_figures"])
for k in d["kernels"]:
    print(k["key"], round(k["avg_launch_us"], 2), "us", round(k["frac_of_hbm_peak"], 3))
for name, leg in r["legs"].items():
    if isinstance(leg, dict):
        print(name, round(leg["avg_launch_us"], 2), "us frac", round(leg["frac"], 3), leg["kernel"][:60])
    else:
        for k in leg:
            print("  n64m", k["key"], round(k["avg_launch_us"], 2), "us", round(k["frac_of_hbm_peak"], 3), k.get("scalars_launch_us"))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1); echo "trace rc=$?" | tee -a $O/status.txt
python3 tools/summarize_prof.py $O/trace > $O/trace_summary.txt 2>&1 || true
python3 tools/kernel_stats_to_json.py $O/trace $O/bench_kernel_stats.json
head -12 $O/trace_summary.txt | cut -c1-200
timeout -k 10 300 python tools/gmres_probe3.py 2000 HIPK_GM_SPLIT_NORM > $O/gmres_split.jsonl 2> $O/gmres_split.err; echo "gmres probe rc=$?" | tee -a $O/status.txt
cut -c1-220 $O/gmres_split.jsonl
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --scaling strong --steps 1 --warmup 0 --no-cpu-baseline > $O/strong_$i.json 2> $O/strong_$i.err; echo "strong $i rc=$?" | tee -a $O/status.txt
  python3 -c "
import json; d = json.load(open('$O/strong_$i.json')); print('strong', d['value'], d['config'].get('placement_probe_GBps'), d['config'].get('placement_allocations_drawn'), d['config']['iterations_per_solve'], d['config']['info'])"
done
