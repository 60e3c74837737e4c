#!/bin/bash
# A/B on one box: tallest strip of k_tvl1_iter3 the height model may pick when contexts share the device (option rows3_max).
cd /root/repo && mkdir -p gpurun_out
python tools/ab_bench.py base=,rows3_max=32 r48=,rows3_max=48 r64=,rows3_max=64 r96=,rows3_max=96 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-cli" > gpurun_out/r04_ab_strip_height.txt 2>&1
tail -6 gpurun_out/r04_ab_strip_height.txt
