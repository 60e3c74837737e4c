#!/bin/bash
# k_tvl1_iter3 with its loads two rows ahead (OFX_ITER3_PF=2) against one: production and memory-ceiling builds, one box.
mkdir -p gpurun_out
python tools/ab_bench.py prod= pf2=variants/libofx_pf2.so mem=variants/libofx_ceil3_mem.so mempf2=variants/libofx_ceil3_mem_pf2.so --rounds 2 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter3_prefetch_depth.txt 2>&1
grep MEDIAN gpurun_out/r04_iter3_prefetch_depth.txt | cut -c1-300
