#!/bin/bash
# round 3 session 14: robust_expo_methods on the GPU (SOR / shim / golden suites), then the whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_sor.py tests/test_gpu_shim.py -m gpu -x -q -k "robust or shim" > $O/rexpo.log 2>&1; echo "rexpo rc=$?"; tail -15 $O/rexpo.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log
python - <<'PY'
import importlib, time, sys
sys.path.insert(0, '.')
ofx = importlib.import_module("optical-flow-1_amd"); synth = importlib.import_module("optical-flow-1_amd.synth")
import oracle
c = ofx.Ofx(0, ofx.F64)
I1, I2 = synth.pair("P0", 1280, 720)
kw = dict(method=1, alpha=50.0, gamma=10.0, lam=0.1, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)
c.robust_expo(I1, I2, **kw)
t0 = time.perf_counter(); u, v = c.robust_expo(I1, I2, **kw); dt = time.perf_counter() - t0
st = c.stats()
r = oracle.Ref(); r.set_num_threads(1)
t0 = time.perf_counter(); ur, vr = r.robust_expo(I1, I2, **kw); t1 = time.perf_counter() - t0
import numpy as np
print("robust_expo 1280x720 method 1: GPU %.3f s (%d sweeps), reference 1 thread %.2f s, max |delta| %.3g" % (dt, int(st.iterations().sum()), t1, max(np.abs(u - ur).max(), np.abs(v - vr).max())))
PY
