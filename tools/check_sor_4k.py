"""One-off check beyond the test suite's sizes: exact Horn-Schunck and Brox solves at 3840x2160 against the compiled
reference (1 thread, ~50 s of CPU).  Measured on MI355X: HS 0.44 s vs 27.9 s, Brox 0.46 s vs 23.4 s, max |diff| = 0.0."""
import sys, os, time, importlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
ofx = importlib.import_module("optical-flow-1_amd"); synth = importlib.import_module("optical-flow-1_amd.synth")
import oracle
ctx = ofx.Ofx(0, ofx.F64)
cpu = oracle.Ref() if oracle.have_ref() else oracle.Oracle()
cpu.set_num_threads(1)
nx, ny = 3840, 2160
I1, I2 = synth.pair("P1", nx, ny, 1)
for name, g, c, kw in [("hs", ctx.hs_pyramidal, cpu.hs_pyramidal, dict(alpha=20.0, nscales=5, zfactor=0.5, warps=3, TOL=1e-4, maxiter=150)),
                       ("brox", ctx.brox_spatial, cpu.brox_spatial, dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=4))]:
    t0 = time.perf_counter(); r = g(I1, I2, **kw); tg = time.perf_counter() - t0
    it_g = ctx.stats().iterations().copy()
    t0 = time.perf_counter(); o = c(I1, I2, **kw); tc = time.perf_counter() - t0
    print(name, "gpu %.2fs cpu %.2fs" % (tg, tc), "max|d|", max(np.abs(r[0] - o[0]).max(), np.abs(r[1] - o[1]).max()),
          "sweeps equal", np.array_equal(np.asarray(it_g).ravel()[:np.asarray(o[2]).size], np.asarray(o[2]).ravel()) if len(o) > 2 else None, flush=True)
