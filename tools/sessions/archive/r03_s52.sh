#!/bin/bash
# round 3 session 52: steady-state step of k_tvl1_iter3 (six stages in one basic block): parity, A/B (3 waves with spills / 2 waves / without)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03az; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3 or strips" > $O/tvl1_tests.log 2>&1; rc=$?; echo "tvl1 tests rc=$rc"; tail -4 $O/tvl1_tests.log
[ $rc -ne 0 ] && exit 1
OFX_LIB_PATH=$R/variants/libofx_steady_w2.so timeout -k 10 600 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3" > $O/tvl1_tests_w2.log 2>&1; rc=$?; echo "tvl1 tests (2 waves) rc=$rc"; tail -2 $O/tvl1_tests_w2.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 1100 python tools/ab_bench.py "nosteady=variants/libofx_nosteady.so" "steady_w3=" "steady_w2=variants/libofx_steady_w2.so" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
