#!/bin/bash
# The 1080p x 5 launch of k_tvl1_iter3 alone (bench.py's roofline leg) at forced strip heights: launch_us is the figure to read.
mkdir -p gpurun_out
python tools/ab_bench.py base= r16=,rows_per_wave3=16 r24=,rows_per_wave3=24 r48=,rows_per_wave3=48 r64=,rows_per_wave3=64 r128=,rows_per_wave3=128 --rounds 1 --args "--no-cpu --no-sor --no-occ --no-4k --no-cli" > gpurun_out/r04_ab_strip_height_launch.txt 2>&1
tail -7 gpurun_out/r04_ab_strip_height_launch.txt
