#!/bin/bash
set -o pipefail
O=gpurun_out/r02c7
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_mode.json 2> $O/bench_mode.err; echo "bench mode rc=$?" | tee -a $O/status.txt
HIPK_SPMV_SELL_NO_MODE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $O/bench_nomode.json 2> $O/bench_nomode.err; echo "bench nomode rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json
for f in ("bench_mode","bench_nomode"):
    d=json.loads(open(f"gpurun_out/r02c7/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), [ (k["key"], round(k["avg_launch_us"],2)) for k in d["kernels"]], round(d["spmv_standalone"]["us"],2))
PY
# PMC evidence for the coded SpMV (separate passes, no trace domains mixed in)
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
HIPK_SPMV_SELL_NO_PAIR=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/pmc_insts -- python3 $ARGS > $O/pmc_insts.log 2>&1
HIPK_SPMV_SELL_NO_PAIR=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_cycles -- python3 $ARGS > $O/pmc_cycles.log 2>&1
HIPK_SPMV_SELL_NO_PAIR=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_l2 -- python3 $ARGS > $O/pmc_l2.log 2>&1
python3 tools/summarize_prof.py $O > $O/pmc_summary.txt 2>&1
grep -A3 "sell_loop\|cg_direction" $O/pmc_summary.txt | head -40
