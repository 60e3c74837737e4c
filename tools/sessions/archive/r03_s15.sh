#!/bin/bash
# round 3 session 15: randomised soak with other draws (all kernel choices at their defaults), incl. tolerance mode and robust_expo
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
OFX_FUZZ_SEED=131 OFX_FUZZ_N=140 OFX_FUZZ_SOR=20 OFX_FUZZ_GROUPS=14 OFX_FUZZ_TEMPORAL=6 OFX_FUZZ_SOR_GROUPS=10 OFX_FUZZ_OCC=20 OFX_FUZZ_REXPO=30 \
  timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz_soak.log 2>&1; echo "soak rc=$?"; tail -4 $O/fuzz_soak.log
