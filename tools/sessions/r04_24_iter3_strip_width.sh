#!/bin/bash
# k_tvl1_iter3 with 56 output columns per wave (stores = whole 128-byte lines, four halo lanes each side) against 58: production and memory ceiling.
mkdir -p gpurun_out
python tools/ab_bench.py prod= s56=variants/libofx_s56.so mem=variants/libofx_ceil3_mem.so mems56=variants/libofx_ceil3_mem_s56.so --rounds 2 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter3_strip_width.txt 2>&1
grep MEDIAN gpurun_out/r04_iter3_strip_width.txt | cut -c1-300
