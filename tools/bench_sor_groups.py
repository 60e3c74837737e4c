#!/usr/bin/env python3
"""BASELINE configs 3 and 4 through the lockstep-group entry points: G pairs per launch on one context, and batches on
several contexts.  Prints one JSON line per (config, contexts, group size): seconds, Mpix*sweeps/s, algorithmic GB/s
(56 B / 80 B per pixel-sweep) and the fraction of the 8 TB/s HBM peak.  --check: every flow of the largest
run is compared bit for bit with the pair solved alone."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
small = "--small" in sys.argv
check = "--check" in sys.argv
only = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--only=")]
dev = torch.device("cuda", 0)
CONFIGS = [("hs_cfg3", (640, 360) if small else (1920, 1080), 56.0,
            dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)),
           ("brox_cfg4", (320, 180) if small else (1280, 720), 80.0,
            dict(alpha=50.0, gamma=10.0, nscales=4 if small else 6, nu=0.5, TOL=1e-4, inner=1, outer=15))]
GRID = [(1, 1), (1, 4), (1, 16), (2, 16), (4, 16)]           # (contexts, pairs per group)
NPAIRS = 16                                                  # every grid point solves the SAME first 16 pairs (x contexts > 1: 16 per context)
for a in sys.argv[1:]:
    if a.startswith("--grid="):                              # e.g. --grid=1x16,2x16
        GRID = [tuple(int(x) for x in g.split("x")) for g in a.split("=")[1].split(",")]
nowarm = "--no-warm" in sys.argv
OPTS = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--opt=")]      # --opt=name=value for every context
ODD = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--opt-odd=")]
KW = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kw=")]         # --kw=name=value overrides a solver parameter
for name, (nx, ny), bpp, kw in CONFIGS:
    if only and name not in only:
        continue
    for o in KW:
        k_, v_ = o.split("=")
        if k_ in kw:
            kw[k_] = type(kw[k_])(float(v_))
    nmax = max(NPAIRS, max(c * g for c, g in GRID))
    ins = [synth.pair_device("P0" if k == 0 else "P1", nx, ny, k, dev) for k in range(nmax)]
    flo = torch.empty((nmax, ny, nx, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    fn = ofx.hs_batch_dev if name.startswith("hs") else ofx.brox_batch_dev
    for nctx, G in GRID:
        ctxs = [ofx.Ofx(0, ofx.F64) for _ in range(nctx)]
        for ci, c in enumerate(ctxs):
            c.set_option("lockstep", G)
            for o in OPTS:
                c.set_option(o.split("=")[0], float(o.split("=")[1]))
            for o in ODD if ci % 2 else []:                   # --opt-odd=name=value: only for every second context
                c.set_option(o.split("=")[0], float(o.split("=")[1]))
        n = max(NPAIRS, nctx * G)
        args = ([t[0].data_ptr() for t in ins[:n]], [t[1].data_ptr() for t in ins[:n]], [flo[k].data_ptr() for k in range(n)], nx, ny)
        if not nowarm:
            fn(ctxs, *args, **kw)                              # warm (arena, snapshots, clocks)
        t0 = time.perf_counter()
        work = fn(ctxs, *args, **kw)
        dt = time.perf_counter() - t0
        mps = sum(work) / dt / 1e6
        print(json.dumps({"config": name, "size": "%dx%d" % (nx, ny), "contexts": nctx, "group": G, "pairs": n,
                          "seconds": round(dt, 4), "mpix_sweeps": round(sum(work) / 1e6, 1), "ms_per_pair": round(dt / n * 1e3, 2), "mpix_sweeps_per_s": round(mps, 1),
                          "algorithmic_gbs": round(mps * bpp / 1e3, 1), "frac_of_hbm_peak": round(mps * bpp / 8e6, 4)}), flush=True)
        if check and (nctx, G) == GRID[-1]:
            solo = ofx.Ofx(0, ofx.F64)
            for o in OPTS:
                solo.set_option(o.split("=")[0], float(o.split("=")[1]))
            one = torch.empty((ny, nx, 2), dtype=torch.float32, device=dev)
            gfn = solo.hs_group_dev if name.startswith("hs") else solo.brox_group_dev
            bad = 0
            for k in range(n):
                gfn([ins[k][0].data_ptr()], [ins[k][1].data_ptr()], [one.data_ptr()], nx, ny, **kw)
                solo.synchronize()
                bad += int(not torch.equal(one.view(torch.int32), flo[k].view(torch.int32)))
            print(json.dumps({"config": name, "check": "%d of %d flows differ from the solo solve" % (bad, n)}), flush=True)
            solo.close()
        for c in ctxs:
            c.close()
