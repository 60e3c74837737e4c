#!/bin/bash
# Tolerance arithmetic with one refinement step on rsq / rcp (default build) against two + correction (variants/libofx_newton2.so):
# the TV-L1 tests of the tolerance / f32 modes, then the job A/B on one box.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tvl1.py -x -q -m gpu -k "tolerance or f32 or float or fast or relaxed" > gpurun_out/r04_pytest_tol_newton1.txt 2>&1
tail -5 gpurun_out/r04_pytest_tol_newton1.txt
python tools/ab_bench.py n1= n2=variants/libofx_newton2.so --rounds 3 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_ab_tolerance_refinement.txt 2>&1
grep MEDIAN gpurun_out/r04_ab_tolerance_refinement.txt
