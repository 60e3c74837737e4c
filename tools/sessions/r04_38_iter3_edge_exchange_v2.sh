#!/bin/bash
# Edge exchange, second form: lane 0 reads the loaded p of its left neighbour itself (no wait for the prefetch before the barrier).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
OFX_FUSE3_XC=1 timeout -k 10 600 python -m pytest tests/test_gpu_tvl1.py -x -q -m gpu > gpurun_out/r04_pytest_xc2.txt 2>&1; tail -2 gpurun_out/r04_pytest_xc2.txt
python tools/ab_bench.py base= xc=,fuse3_xc=1 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_ab_iter3_edge_exchange_v2.txt 2>&1
grep MEDIAN gpurun_out/r04_ab_iter3_edge_exchange_v2.txt | cut -c1-400
