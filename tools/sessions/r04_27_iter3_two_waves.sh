#!/bin/bash
# k_tvl1_iter3 (56 columns) compiled for 2 waves per SIMD against 3: production and ALU ceiling -- what a four-iteration kernel (~205 VGPRs) could expect.
mkdir -p gpurun_out
python tools/ab_bench.py prod= w2=variants/libofx_w2.so alu=variants/libofx_ceil3_alu.so aluw2=variants/libofx_ceil3_alu_w2.so --rounds 2 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter3_two_waves.txt 2>&1
grep MEDIAN gpurun_out/r04_iter3_two_waves.txt | cut -c1-400
