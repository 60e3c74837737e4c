#!/usr/bin/env python3
"""Sweep of the windowed-SOR launch geometry with a lockstep group of 16 pairs on one context (cfg 3 / cfg 4):
rows per workgroup (61 = one wave per workgroup incl. the three border items, 64 = two waves) x steps per launch."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
dev = torch.device("cuda", 0)
G = 16
CONFIGS = [("hs_cfg3", (1920, 1080), ofx.hs_batch_dev, dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)),
           ("brox_cfg4", (1280, 720), ofx.brox_batch_dev, dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15))]
for name, (nx, ny), fn, kw in CONFIGS:
    ins = [synth.pair_device("P0" if k == 0 else "P1", nx, ny, k, dev) for k in range(G)]
    flo = torch.empty((G, ny, nx, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx = ofx.Ofx(0, ofx.F64)
    ctx.set_option("lockstep", G)
    args = ([t[0].data_ptr() for t in ins], [t[1].data_ptr() for t in ins], [flo[k].data_ptr() for k in range(G)], nx, ny)
    for rows, window, spw in [(r_, w_, 1) for r_ in (61, 64, 125, 29) for w_ in (4, 8, 16)] if "--geometry" in sys.argv else \
            [(125, w_, p_) for p_ in (1, 2, 4) for w_ in (4, 8)] + [(61, 8, 2), (253, 8, 1), (253, 8, 2)]:
        if True:
            ctx.set_option("sor_rows", rows)
            ctx.set_option("sor_window", window)
            ctx.set_option("sor_spw", spw)
            fn([ctx], *args, **kw)
            t0 = time.perf_counter()
            work = fn([ctx], *args, **kw)
            dt = time.perf_counter() - t0
            print(json.dumps({"config": name, "group": G, "sor_rows": rows, "sor_window": window, "sor_spw": spw, "seconds": round(dt, 3),
                              "mpix_sweeps_per_s": round(sum(work) / dt / 1e6, 1)}), flush=True)
    ctx.close()
