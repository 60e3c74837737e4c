#!/bin/bash
# round 3 session 36: wave index made uniform for the compiler (readfirstlane) and the branch-light primal stage: parity, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03aj; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3 or strips" > $O/tvl1_tests.log 2>&1; rc=$?; echo "tvl1 tests rc=$rc"; tail -5 $O/tvl1_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 1100 python tools/ab_bench.py "before=variants/libofx_norfl.so" "uniform=variants/libofx_rfl_only.so" "uniform_fastprimal=" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
