#!/bin/bash
# round 3 session 38: where a batch of 32 640x480 triples spends its time (kernel trace: GPU busy share, per-stream gaps)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/occb -- python3 $R/tools/bench_tvl1occ.py --size 640x480 --cpu none --batch 2:32 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
grep '"size"' $O/trace.log | cut -c1-300
cd $R
python3 tools/fmt_kernel_stats.py /tmp/occb 14 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
python3 tools/trace_budget.py /tmp/occb --all > $O/budget.txt 2>&1; grep -A12 "per stream" $O/budget.txt | cut -c1-220; head -3 $O/budget.txt
ls /tmp/occb/*/ | head; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/occb/*/*memory_copy_trace.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot={}
    for r in rows:
        k=r.get('Direction') or r.get('Name')
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        t=tot.setdefault(k,[0,0.0]); t[0]+=1; t[1]+=d
    print({k:(v[0],round(v[1],2)) for k,v in tot.items()})
PY
