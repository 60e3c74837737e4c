#!/usr/bin/env python3
"""Brox temporal on a synthetic 640x480 sequence: GPU (exact windowed schedule) next to the compiled
reference on the host cores.  Prints one JSON line."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
import oracle

nx, ny, frames = 640, 480, 5
kw = dict(nscales=4, nu=0.75, outer=15, inner=1)
I = synth.sequence(nx, ny, frames)
G = ofx.Ofx(0, ofx.F64)
cpu = oracle.Ref() if oracle.have_ref() else oracle.Oracle()
cores = min(oracle.host_cores(), 32)
G.brox_temporal(I, **kw)                                   # warm (arena, clocks)
t0 = time.perf_counter()
ug, vg = G.brox_temporal(I, **kw)
tg = time.perf_counter() - t0
st = G.stats()
cpu.set_num_threads(1)
t0 = time.perf_counter()
r = cpu.brox_temporal(I, **kw)
t1 = time.perf_counter() - t0
cpu.set_num_threads(cores)
t0 = time.perf_counter()
cpu.brox_temporal(I, **kw)
tn = time.perf_counter() - t0
print(json.dumps({"config": "brox_temporal %dx%d x %d frames, %s" % (nx, ny, frames, kw),
                  "gpu_exact": {"seconds": round(tg, 4), "sweeps": int(st.iterations().sum()),
                                "mpix_sweeps_per_s": round(st.work_pix_iters / tg / 1e6, 1),
                                "max_abs_diff_vs_reference_1thread": float(max(np.abs(ug - r[0]).max(), np.abs(vg - r[1]).max()))},
                  "cpu_reference": {"kind": cpu.kind, "seconds_1_thread": round(t1, 3), "seconds_%d_threads" % cores: round(tn, 3)}}))
