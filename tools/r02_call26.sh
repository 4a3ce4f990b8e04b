#!/bin/bash
# round 2, call 26: final selection rule of the strided walk (tests), counters of the N = 64 M SpMV (chunk walk vs strided walk),
# the row-partitioned rehearsal shapes again
set -o pipefail
O=gpurun_out/r02c26
mkdir -p $O
export TMPDIR=/tmp
export PYTHONPATH=$PWD/pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd:$PYTHONPATH
timeout -k 10 400 python -m pytest tests/test_gpu_coded.py -m gpu -x -q -k "two_rows or many_grid_lines" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_new.log
grep -q "pytest new rc=0" $O/status.txt || exit 1
timeout -k 10 300 python tools/walk_probe.py 4000 5657 8000 > $O/walk.log 2>&1; echo "walk rc=$?" | tee -a $O/status.txt
grep -v Warning $O/walk.log | cut -c1-360
grep -q "walk rc=0" $O/status.txt || exit 1
export WALK_SETTINGS="0:,1:1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/walk_probe.py 8000 > $O/trace.log 2>&1; echo "trace rc=$?" | tee -a $O/status.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/walk_probe.py 8000 > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?" | tee -a $O/status.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/walk_probe.py 8000 > $O/pmc_write.log 2>&1; echo "write rc=$?" | tee -a $O/status.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_l2 -- python3 tools/walk_probe.py 8000 > $O/pmc_l2.log 2>&1; echo "l2 rc=$?" | tee -a $O/status.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq -- python3 tools/walk_probe.py 8000 > $O/pmc_sq.log 2>&1; echo "sq rc=$?" | tee -a $O/status.txt
unset WALK_SETTINGS
python3 tools/pmc_kernels.py $O hipk_spmv_sell hipk_cg_ > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | cut -c1-420
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" > $O/kernel_stats.csv && cat $O/kernel_stats.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
