#!/bin/bash
# round 2, call 33: whole GPU suite + smoke on the final code
set -o pipefail
O=gpurun_out/r02c33
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_all.log | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
tail -1 $O/smoke.log
