#!/bin/bash
# round 2, GPU call 2: GPU suite on the new code (transpose handle, config-4 fixtures, sweep GMRES), GMRES(30) at N=4M A/B
set -o pipefail
O=gpurun_out/r02c2
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
for i in 1 2; do
python tools/gmres_probe.py 2000 batched >> $O/gm_sweep.log 2>&1
HIPK_GMRES_NO_SWEEP=1 python tools/gmres_probe.py 2000 batched >> $O/gm_legacy.log 2>&1
done
python tools/gmres_probe.py 2000 incremental >> $O/gm_sweep.log 2>&1
grep cycles $O/gm_sweep.log $O/gm_legacy.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gm_trace -- python3 tools/gmres_probe.py 2000 batched > $O/gm_trace.log 2>&1
python3 tools/summarize_prof.py $O/gm_trace > $O/gm_trace_summary.txt 2>&1 || true
head -30 $O/gm_trace_summary.txt
