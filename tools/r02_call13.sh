#!/bin/bash
set -o pipefail
O=gpurun_out/r02c13
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_pcg.py tests/test_distributed_gloo.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest.log
grep -q "rc=0" $O/status.txt || exit 1
timeout -k 10 300 python tools/gmres_variants.py 2000 HIPK_GMRES_NO_STREAM=1,HIPK_GM_SPEC=0 HIPK_GM_SPEC=0 "" HIPK_GM_NRES=31 2>&1 | grep cycle | tee $O/variants.log
timeout -k 10 500 bash tools/prof_solvers.sh r02c13/prof_solvers > $O/prof_solvers.log 2>&1; echo "prof_solvers rc=$?" | tee -a $O/status.txt
python tools/pmc_to_json.py $O/prof_solvers $O/pmc_solvers.json --commit "$(cat .commit_stamp 2>/dev/null)" > /dev/null 2>&1; echo "pmc solvers json rc=$?" | tee -a $O/status.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/prof_solvers/gmres/summary.txt | head -12
