#!/bin/bash
# one-line-per-setting summary of a tools/walk_probe.py log and the PMC summary beside it (dev aid)
O=$1
cat $O/status.txt; tail -1 $O/pytest_new.log
grep -v Warn $O/walk.log | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['nx'], 'st',d['strided'],'nt',d['nt_y'],'walk',d['walk'], d['kernel'][-20:], 'alone',d['spmv_alone_us'],'incg',d['spmv_in_cg_us'],'cg',d['cg_us_per_iter'])
    elif l.startswith('nx'): print(l)
"
grep "spmv_sell" $O/pmc_summary.txt | python3 -c "
import sys,re
for l in sys.stdin:
    name=l.split('(')[0]
    g=lambda k: re.search(\"'%s': ([0-9.]+)\"%k,l)
    print(name[-36:], 'FETCH_MB', round(float(g('FETCH_SIZE').group(1))*2*1024/1e6), 'WRITE_MB', round(float(g('WRITE_SIZE').group(1))*1024/1e6), 'hit', g('TCC_HIT_sum').group(1),'miss',g('TCC_MISS_sum').group(1),'rdreq',g('TCP_TCC_READ_REQ_sum').group(1), 'wavecyc', g('SQ_WAVE_CYCLES').group(1), 'wait', g('SQ_WAIT_INST_ANY').group(1))
"
grep spmv_sell $O/kernel_stats.csv | cut -c1-120
