"""One-off check beyond the test suite's sizes (BASELINE config 5 size): TV-L1 and the exact Horn-Schunck and Brox
solves at 3840x2160 against the compiled reference (1 thread, ~90 s of CPU).  Measured on MI355X: TV-L1 0.05 s (host buffers in / out) vs 3.4 s on 16
threads, iteration tables equal; HS 0.42 s vs 27.7 s, Brox 0.47 s vs 23.6 s; max |diff| = 0.0 in all three."""
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
t0 = time.perf_counter(); ug, vg = ctx.tvl1_multiscale(I1, I2); tg = time.perf_counter() - t0
it_g = ctx.stats().iterations().copy()
port = oracle.Oracle()                       # the restatement reports the iteration table; TV-L1 has no racy loop
port.set_num_threads(min(oracle.host_cores(), 16))
t0 = time.perf_counter(); uo, vo, it_o, _ = port.tvl1_multiscale(I1, I2); tc = time.perf_counter() - t0
port.set_num_threads(1)
print("tvl1 gpu %.2fs cpu %.2fs" % (tg, tc), "max|d|", max(np.abs(ug - uo).max(), np.abs(vg - vo).max()),
      "iteration tables equal", bool(np.array_equal(it_g, it_o)), flush=True)
for name, g, c, kw in [("hs", ctx.hs_pyramidal, cpu.hs_pyramidal, dict(alpha=20.0, nscales=5, zfactor=0.5, warps=3, TOL=1e-4, maxiter=150)),
                       ("brox", ctx.brox_spatial, cpu.brox_spatial, dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=4))]:
    t0 = time.perf_counter(); r = g(I1, I2, **kw); tg = time.perf_counter() - t0
    it_g = ctx.stats().iterations().copy()
    t0 = time.perf_counter(); o = c(I1, I2, **kw); tc = time.perf_counter() - t0
    print(name, "gpu %.2fs cpu %.2fs" % (tg, tc), "max|d|", max(np.abs(r[0] - o[0]).max(), np.abs(r[1] - o[1]).max()),
          "sweeps equal", np.array_equal(np.asarray(it_g).ravel()[:np.asarray(o[2]).size], np.asarray(o[2]).ravel()) if len(o) > 2 else None, flush=True)
