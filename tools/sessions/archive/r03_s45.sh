#!/bin/bash
# round 3 session 45: randomised soak of the final tree with other draws (all defaults), then with the alternative schedules forced
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03as; mkdir -p $O
cd $R
OFX_FUZZ_SEED=307 OFX_FUZZ_N=150 OFX_FUZZ_SOR=24 OFX_FUZZ_GROUPS=16 OFX_FUZZ_TEMPORAL=6 OFX_FUZZ_SOR_GROUPS=12 OFX_FUZZ_OCC=24 OFX_FUZZ_REXPO=24 \
  timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/soak_defaults.log 2>&1; echo "soak defaults rc=$?"; tail -2 $O/soak_defaults.log
OFX_FUZZ_OPTS="fuse3=1,rof_window=24,sor_lds=2" OFX_FUZZ_SEED=308 OFX_FUZZ_N=80 OFX_FUZZ_SOR=16 OFX_FUZZ_GROUPS=12 OFX_FUZZ_TEMPORAL=2 OFX_FUZZ_SOR_GROUPS=6 OFX_FUZZ_OCC=16 OFX_FUZZ_REXPO=4 \
  timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/soak_alt.log 2>&1; echo "soak alternatives rc=$?"; tail -2 $O/soak_alt.log
