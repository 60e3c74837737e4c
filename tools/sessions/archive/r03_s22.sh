#!/bin/bash
# round 3 session 22: batches of 640x480 triples with the iterations in flight -- group size x contexts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
for spec in "1:16 16" "2:32 16" "2:32 8" "4:32 8" "4:32 4" "4:64 16" "2:32 16 rof_pipe=0" "4:32 8 rof_pipe=0"; do
  set -- $spec
  extra=""; [ -n "$3" ] && extra="--opt $3"
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --cpu none --batch $1 --opt lockstep=$2 $extra 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'options': d['options'], 'one_triple_s': d['gpu_s'], 'batch': d['batch']}))" || exit 1
done | tee $O/occ_batches.txt
