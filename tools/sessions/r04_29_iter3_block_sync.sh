#!/bin/bash
# k_tvl1_iter3: a workgroup barrier per marching step (adjacent strips read their shared halo lines together) against none: production and memory ceiling.
mkdir -p gpurun_out
python tools/ab_bench.py prod= sync=variants/libofx_sync.so mem=variants/libofx_ceil3_mem.so memsync=variants/libofx_ceil3_mem_sync.so --rounds 3 --args "--no-cpu --no-sor --no-occ --no-cli --no-4k" > gpurun_out/r04_iter3_block_sync.txt 2>&1
grep MEDIAN gpurun_out/r04_iter3_block_sync.txt | cut -c1-400
