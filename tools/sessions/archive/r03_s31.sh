#!/bin/bash
# round 3 session 31: radius-templated fused Gaussian against the generic one, interleaved on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ae; mkdir -p $O
cd $R
timeout -k 10 1100 python tools/ab_bench.py "gauss_generic=,gauss_fused=2" "gauss_templated=" --rounds 4 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
