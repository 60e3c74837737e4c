cd $GRAFT_REPO_ROOT
echo "== soak 1: every option at its default"
OFX_FUZZ_SEED=401 OFX_FUZZ_N=150 OFX_FUZZ_SOR=24 OFX_FUZZ_GROUPS=16 OFX_FUZZ_TEMPORAL=6 OFX_FUZZ_SOR_GROUPS=12 OFX_FUZZ_OCC=24 OFX_FUZZ_REXPO=24 OFX_FUZZ_SOR_TOL=60 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -4
echo "== soak 2: three iterations per launch at every size (cursor loop), LDS windows, 24-step ROF windows"
OFX_FUZZ_OPTS="fuse3=1,rof_window=24,sor_lds=2" OFX_FUZZ_SEED=402 OFX_FUZZ_N=120 OFX_FUZZ_SOR=12 OFX_FUZZ_GROUPS=24 OFX_FUZZ_TEMPORAL=2 OFX_FUZZ_SOR_GROUPS=8 OFX_FUZZ_OCC=10 OFX_FUZZ_REXPO=6 OFX_FUZZ_SOR_TOL=30 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -4
echo "== soak 3: cursor loop with single-iteration units everywhere"
OFX_FUZZ_OPTS="fuse3=1,fuse3_afac1=1e9" OFX_FUZZ_SEED=403 OFX_FUZZ_N=80 OFX_FUZZ_SOR=0 OFX_FUZZ_GROUPS=24 OFX_FUZZ_TEMPORAL=0 OFX_FUZZ_SOR_GROUPS=0 OFX_FUZZ_OCC=0 OFX_FUZZ_REXPO=0 OFX_FUZZ_SOR_TOL=0 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -4
