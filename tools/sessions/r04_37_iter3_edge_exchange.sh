#!/bin/bash
# k_tvl1_iter3 with the four waves of a workgroup exchanging their edge columns through LDS (option fuse3_xc / OFX_FUSE3_XC=1): parity, then A/B on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
OFX_FUSE3_XC=1 timeout -k 10 600 python -m pytest tests/test_gpu_tvl1.py -x -q -m gpu > gpurun_out/r04_pytest_xc.txt 2>&1; tail -4 gpurun_out/r04_pytest_xc.txt
OFX_FUSE3_XC=1 OFX_FUZZ_OPTS="fuse3=1" OFX_FUZZ_SEED=411 OFX_FUZZ_N=120 OFX_FUZZ_SOR=0 OFX_FUZZ_GROUPS=24 OFX_FUZZ_TEMPORAL=0 OFX_FUZZ_SOR_GROUPS=0 OFX_FUZZ_OCC=0 OFX_FUZZ_REXPO=0 OFX_FUZZ_SOR_TOL=0 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04_pytest_xc_fuzz.txt 2>&1; tail -3 gpurun_out/r04_pytest_xc_fuzz.txt
python tools/ab_bench.py base= xc=,fuse3_xc=1 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-cli" > gpurun_out/r04_ab_iter3_edge_exchange.txt 2>&1
grep MEDIAN gpurun_out/r04_ab_iter3_edge_exchange.txt | cut -c1-400
