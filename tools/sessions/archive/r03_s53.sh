#!/bin/bash
# round 3 session 53: steady-state step of k_tvl1_iter3 at 2 waves per SIMD (no scratch): primal stages / dual stages as two scheduling regions (1),
# every stage its own region (2), fully interleaved (0); against the production kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ba; mkdir -p $O
cd $R
for v in st1w2 st2w2 st0w2; do
  OFX_LIB_PATH=$R/variants/libofx_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3" > $O/tests_$v.log 2>&1; rc=$?; echo "tests $v rc=$rc"; tail -1 $O/tests_$v.log
  [ $rc -ne 0 ] && exit 1
done
timeout -k 10 1100 python tools/ab_bench.py "production=" "two_regions=variants/libofx_st1w2.so" "six_regions=variants/libofx_st2w2.so" "one_region=variants/libofx_st0w2.so" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
