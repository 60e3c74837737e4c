#!/usr/bin/env python3
"""Exact SOR solvers with several independent pairs in flight on ONE GPU: N contexts (HIP stream + workspace
each), one host thread per context, every thread solving BASELINE config 3 (Horn-Schunck 1920x1080) or config 4
(Brox 1280x720) on its own pair.  An exact solve is a latency chain that occupies a small part of the chip, so
independent pairs overlap.  Prints one JSON line per (config, N).

usage: bench_sor_concurrent.py [--n=1,2,4,8] [--small]"""
import importlib, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")

ns = [1, 2, 4, 8]
small = "--small" in sys.argv
for a in sys.argv[1:]:
    if a.startswith("--n="):
        ns = [int(x) for x in a[4:].split(",")]

CONFIGS = [
    ("cfg3 horn_schunck_pyramidal", "hs_pyramidal", (640, 360) if small else (1920, 1080), "P0",
     dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)),
    ("cfg4 brox_spatial", "brox_spatial", (320, 180) if small else (1280, 720), "P0",
     dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)),
]

ctxs = [ofx.Ofx(0, ofx.F64) for _ in range(max(ns))]
for name, fn, (nx, ny), pair, kw in CONFIGS:
    # different pairs per context (P1 batch variants for k > 0), so the solves do not march in step
    pairs = [synth.pair(pair, nx, ny)] + [synth.pair("P1", nx, ny, k) for k in range(1, max(ns))]
    solo, solo_s = [], []
    for k in range(max(ns)):                                    # every pair alone on its context (second run timed)
        for rep in range(2):
            t0 = time.perf_counter()
            r = getattr(ctxs[k], fn)(pairs[k][0], pairs[k][1], **kw)
            dt = time.perf_counter() - t0
        solo.append(r)
        solo_s.append(dt)
    for n in ns:
        out = [None] * n

        def work(k):
            out[k] = getattr(ctxs[k], fn)(pairs[k][0], pairs[k][1], **kw)

        th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        same = all(np.array_equal(out[k][0], solo[k][0]) and np.array_equal(out[k][1], solo[k][1]) for k in range(n))
        print(json.dumps({"config": name, "size": "%dx%d" % (nx, ny), "pairs_in_flight": n, "seconds": round(dt, 4),
                          "one_after_another_seconds": round(sum(solo_s[:n]), 4),
                          "speedup": round(sum(solo_s[:n]) / dt, 2), "identical_to_solo": bool(same)}), flush=True)
