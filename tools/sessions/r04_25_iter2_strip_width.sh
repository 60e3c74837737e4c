#!/bin/bash
# k_tvl1_iter2 with 56 output columns per wave (whole-line stores) against 60; the default build now has 56 columns in k_tvl1_iter3.
mkdir -p gpurun_out
python tools/ab_bench.py prod= s2=variants/libofx_s2_56.so --rounds 3 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter2_strip_width.txt 2>&1
grep MEDIAN gpurun_out/r04_iter2_strip_width.txt | cut -c1-400
