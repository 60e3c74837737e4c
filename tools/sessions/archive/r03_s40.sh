#!/bin/bash
# round 3 session 40: ROF window length sweep (lone solves at three sizes; batch of 32)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03an; mkdir -p $O
cd $R
for K in 24 16 12 10 8 6; do
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --cpu none --opt rof_window=$K 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
print('K', $K, [(json.loads(l)['size'], json.loads(l)['gpu_s']) for l in sys.stdin])" || exit 1
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --cpu none --batch 2:32 --opt lockstep=16 --opt rof_window=$K 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('K', $K, 'batch', d['batch'])" || exit 1
done | tee $O/window_sweep.txt
timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --check --opt rof_window=8 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $O/window_sweep.txt
