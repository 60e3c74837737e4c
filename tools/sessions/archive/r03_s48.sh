#!/bin/bash
# round 3 session 48: strip height of the three-iteration kernel (model's choice against forced heights), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03av; mkdir -p $O
cd $R
timeout -k 10 1100 python tools/ab_bench.py "model=" "r32=,rows_per_wave3=32" "r40=,rows_per_wave3=40" "r48=,rows_per_wave3=48" "r64=,rows_per_wave3=64" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single --fixed-steps 1" 2>&1 | tee $O/ab.txt
