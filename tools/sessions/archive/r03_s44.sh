#!/bin/bash
# round 3 session 44: p = 0 told to the first launch unit instead of filled: parity (TV-L1 file, all kernel modes), A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ar; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_tvl1.py tests/test_gpu_golden_cli.py tests/test_gpu_shim.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python tools/ab_bench.py "filled=,pzero=0" "told=" --rounds 4 --args "--no-cpu --no-sor --no-occ --no-4k --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
