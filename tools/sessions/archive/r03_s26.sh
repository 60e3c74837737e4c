#!/bin/bash
# round 3 session 26: A/B on one box, interleaved: two iterations per launch / three (3 waves per SIMD, 168 VGPRs + 3 dwords of scratch) / three (2 waves per SIMD, no spill)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03z; mkdir -p $O
cd $R
timeout -k 10 1100 python tools/ab_bench.py "iter2=" "iter3=,fuse3=1,fuse3_min_px=500000" "iter3_w2=variants/libofx_i3w2.so,fuse3=1,fuse3_min_px=500000" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single" 2>&1 | tee $O/ab.txt
