#!/bin/bash
# ALU and memory ceilings of k_tvl1_iter3 (tolerance mode), read from bench.py's roofline legs (launch_us: 1080p x 5 alone; fixed: fixed-work job).
mkdir -p gpurun_out
python tools/ab_bench.py prod= alu=variants/libofx_ceil3_alu.so mem=variants/libofx_ceil3_mem.so --rounds 2 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter3_ceilings.txt 2>&1
cat gpurun_out/r04_iter3_ceilings.txt | cut -c1-400
