#!/bin/bash
# round 2, call 42: the direction step as a scalars launch + flat grid (streaming policy): parity, then in-process A/B
set -o pipefail
O=gpurun_out/r02c42
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coded.py -m gpu -x -q -k "multi_step_chunks or config5 or many_grid_lines or n4m_headline" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest.log | cut -c1-300
grep -q "pytest rc=0" $O/status.txt || exit 1
timeout -k 10 400 python tools/flat_probe.py 4000 5657 8000 2>/dev/null | grep "^{" | tee $O/flat.log
