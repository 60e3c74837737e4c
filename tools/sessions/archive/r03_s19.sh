#!/bin/bash
# round 3 session 19: ROF box sweeps with all iterations of a call in flight (rof_pipe): parity, then one triple and batches, both schedules
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_occ.py -m gpu -x -q > $O/occ_tests.log 2>&1; rc=$?; echo "occ tests rc=$rc"; tail -15 $O/occ_tests.log
[ $rc -ne 0 ] && exit 1
for p in 1 0; do
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --size 1920x1080 --cpu none --opt rof_pipe=$p 2>&1 | grep -v amdgpu.ids || exit 1
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --cpu none --batch 2:32 --opt rof_pipe=$p 2>&1 | grep -v amdgpu.ids || exit 1
done | tee $O/occ_bench.txt
timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --check 2>&1 | grep -v amdgpu.ids | tee -a $O/occ_bench.txt
