#!/bin/bash
# round 3 session 21: shorter pipeline lags on one-block levels + fused chi iterations: parity, randomised occ cases, one triple and batches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_occ.py -m gpu -x -q > $O/occ_tests.log 2>&1; rc=$?; echo "occ tests rc=$rc"; tail -15 $O/occ_tests.log
[ $rc -ne 0 ] && exit 1
OFX_FUZZ_SEED=77 OFX_FUZZ_OCC=40 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "occ" > $O/occ_fuzz.log 2>&1; rc=$?; echo "occ fuzz rc=$rc"; tail -5 $O/occ_fuzz.log
[ $rc -ne 0 ] && exit 1
for opts in "" "--opt chi_fuse=0" "--opt rof_pipe=0 --opt chi_fuse=0"; do
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --cpu none $opts 2>&1 | grep -v amdgpu.ids || exit 1
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --cpu none --batch 2:32 $opts 2>&1 | grep -v amdgpu.ids || exit 1
done | tee $O/occ_bench.txt
timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --size 1920x1080 --check 2>&1 | grep -v amdgpu.ids | tee -a $O/occ_bench.txt
