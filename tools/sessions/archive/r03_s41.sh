#!/bin/bash
# round 3 session 41: 10-step ROF windows as the default: occ parity (4 schedules), randomised groups, golden CLI, one triple + check
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ao; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_occ.py tests/test_gpu_golden_cli.py -m gpu -x -q > $O/occ_tests.log 2>&1; rc=$?; echo "occ tests rc=$rc"; tail -4 $O/occ_tests.log
[ $rc -ne 0 ] && exit 1
OFX_FUZZ_SEED=80 OFX_FUZZ_OCC=40 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "occ" > $O/occ_fuzz.log 2>&1; rc=$?; echo "occ fuzz rc=$rc"; tail -3 $O/occ_fuzz.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --check --batch 2:32 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee $O/occ_bench.txt
