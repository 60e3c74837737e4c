#!/bin/bash
# Row loads of the marching kernels with the non-temporal policy against plain loads: production and memory ceiling, one box.
mkdir -p gpurun_out
python tools/ab_bench.py prod= ldnt=variants/libofx_ldnt.so mem=variants/libofx_ceil3_mem.so memldnt=variants/libofx_ceil3_mem_ldnt.so --rounds 2 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_iter3_nt_loads.txt 2>&1
grep MEDIAN gpurun_out/r04_iter3_nt_loads.txt | cut -c1-400
