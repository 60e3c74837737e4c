#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02k
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_occ.py tests/test_abi.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
python - <<'PY'
import importlib, sys, time, numpy as np
sys.path.insert(0, ".")
ofx = importlib.import_module("optical-flow-1_amd")
import oracle
c = ofx.Ofx(0, ofx.F64); o = oracle.Ref() if oracle.have_ref() else oracle.Oracle()
rng = np.random.default_rng(1)
for nx, ny in ((640, 480), (1920, 1080)):
    v1, v2 = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx)); chi = rng.random((ny, nx)); g = 1 / (1 + rng.random((ny, nx)))
    c.occ_solver_u(v1, v2, chi, g, 0.3, 0.15)
    t0 = time.perf_counter(); c.occ_solver_u(v1, v2, chi, g, 0.3, 0.15); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); o.occ_solver_u(v1, v2, chi, g, 0.3, 0.15); tc = time.perf_counter() - t0
    print("Solver_wrt_u %dx%d: GPU %.3f s, %s (1 thread) %.3f s" % (nx, ny, tg, o.kind, tc), flush=True)
PY
