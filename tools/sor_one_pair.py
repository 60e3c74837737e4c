#!/usr/bin/env python3
"""One pair of BASELINE config 3 (Horn-Schunck 1920x1080) or 4 (Brox 1280x720) in a given SOR mode, repeated; for rocprofv3.
Usage: sor_one_pair.py hs|brox [--exact=0|1] [--fuse=K] [--reps=N] [--opt name=value ...]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
which = sys.argv[1] if len(sys.argv) > 1 else "hs"
opts = {"sor_exact": 0}
reps = 5
for a in sys.argv[2:]:
    if a.startswith("--exact="): opts["sor_exact"] = int(a.split("=")[1])
    elif a.startswith("--fuse="): opts["sor_fuse"] = int(a.split("=")[1])
    elif a.startswith("--reps="): reps = int(a.split("=")[1])
    elif a.startswith("--opt"): pass
    elif "=" in a:
        k, v = a.split("=")
        opts[k] = float(v)
ctx = ofx.Ofx(0, ofx.F64)
for k, v in opts.items():
    ctx.set_option(k, v)
if which == "hs":
    I1, I2 = synth.pair("P0", 1920, 1080)
    fn = lambda: ctx.hs_pyramidal(I1, I2, alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
else:
    I1, I2 = synth.pair("P0", 1280, 720)
    fn = lambda: ctx.brox_spatial(I1, I2, alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)
fn()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    fn()
    ts.append(time.perf_counter() - t0)
st = ctx.stats()
print(json.dumps({"which": which, "options": opts, "seconds_min": round(min(ts), 4), "seconds": [round(t, 4) for t in ts],
                  "sweeps": int(st.iterations().sum()), "per_level": [int(x) for x in st.iterations().sum(axis=1)],
                  "mpix_sweeps_per_s": round(st.work_pix_iters / min(ts) / 1e6, 1)}))
