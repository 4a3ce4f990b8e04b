#!/bin/bash
# round 3, call 14: bench.py with the committed profile digests in place (rocprofv3_avg_us / traffic quoted when the build matches),
# the N > 1 branch at world 1 through the public entry (HIPK_BENCH_DIST=1), and the new GPU tests added after the evidence run
set -o pipefail
O=gpurun_out/r03c14
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_bench_launch.py -m gpu -x -q -k "pytree or bench_line" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log | cut -c1-200
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dist_world1.json 2> $O/bench_dist.err; echo "bench dist rc=$?" | tee -a $O/status.txt
HIPK_BENCH_DIST=1 HIPK_DIST_COMM=fused timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dist_world1_fused.json 2> $O/bench_dist_fused.err; echo "bench dist fused rc=$?" | tee -a $O/status.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c14/bench.json"))
r = d["roofline"]
print("value", round(d["value"], 1), "frac", round(r["frac"], 3), "frac_rocprofv3", r["frac_rocprofv3"], "traffic", r["traffic"], r["traffic_dropped_because"])
for name, leg in r["legs"].items():
    if isinstance(leg, dict):
        print(name, round(leg["avg_launch_us"], 2), round(leg["frac"], 3), leg["rocprofv3_avg_us"], leg["frac_rocprofv3"], leg["traffic"])
for f in ("bench_dist_world1", "bench_dist_world1_fused"):
    x = json.load(open(f"gpurun_out/r03c14/{f}.json"))
    print(f, round(x["value"], 1), x["config"]["collectives"], x["config"]["step"], x["config"]["info"])
PY
