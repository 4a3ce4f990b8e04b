#!/bin/bash
# round 3, final evidence run (one gpurun call): full GPU suite, bench.py, rocprofv3 --kernel-trace --stats of the bench command,
# the FETCH_SIZE / WRITE_SIZE counter passes (separate runs, kernel trace only), digests stamped with the library's build id
# usage (GPU box, repo root): bash tools/r03_final.sh [outdir]
set -o pipefail
O=${1:-gpurun_out/r03final}
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log | cut -c1-200
grep -q "pytest rc=0" $O/status.txt || { grep -n "Error\|assert\|rror:" $O/pytest.log | head -30; exit 1; }
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
grep -q "bench rc=0" $O/status.txt || { tail -20 $O/bench.err; exit 1; }
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $ARGS > $R/$O/trace.log 2>&1); echo "trace rc=$?" | tee -a $O/status.txt
python3 tools/summarize_prof.py $O/trace > $O/trace_summary.txt 2>&1 || true
python3 tools/kernel_stats_to_json.py $O/trace $O/bench_kernel_stats.json | tee -a $O/status.txt
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof/pmc_fetch -- python3 $ARGS > $R/$O/pmc_fetch.log 2>&1); echo "pmc fetch rc=$?" | tee -a $O/status.txt
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof/pmc_write -- python3 $ARGS > $R/$O/pmc_write.log 2>&1); echo "pmc write rc=$?" | tee -a $O/status.txt
python3 tools/pmc_to_json.py $O/prof $O/pmc_kernels.json > $O/pmc_to_json.log 2>&1; echo "pmc json rc=$?" | tee -a $O/status.txt
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print("value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 2), "cold", round(d["config"]["cold_first_solve_ms"], 1))
for name, leg in r["legs"].items():
    if isinstance(leg, dict):
        print(name, round(leg["avg_launch_us"], 2), "us frac", round(leg["frac"], 3))
p = json.load(open("$O/pmc_kernels.json"))
print("pmc build", p.get("build_id"), {k: v["traffic_bytes_per_launch"] for k, v in p["kernels"].items()}, {k: v["traffic_bytes_per_launch"] for k, v in p.get("kernels_n64m", {}).items()})
PY
# the large CSV trees stay on the box: only the digests travel back
rm -rf $O/prof $O/trace/*/*kernel_trace.csv
