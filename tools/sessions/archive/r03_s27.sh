#!/bin/bash
# round 3 session 27: iter3 with register rings: parity (TV-L1 suite incl. the new tolerance-mode headline test), A/B against the plain march
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03aa; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3 or strips" > $O/tvl1_tests.log 2>&1; rc=$?; echo "tvl1 tests rc=$rc"; tail -6 $O/tvl1_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 1100 python tools/ab_bench.py "iter2=,fuse3=0" "ring=" "noring=variants/libofx_noring.so" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single" 2>&1 | tee $O/ab.txt
