#!/usr/bin/env python3
"""Sweep the strip height (rows_per_wave) of k_tvl1_iter at one image size: average launch time from HIP events."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
prec = ofx.F32 if "--f32" in sys.argv else ofx.F64
sizes = [(3840, 2160), (1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]
ctx = ofx.Ofx(0, prec)
ctx.set_option("profile", 1)
rng = np.random.default_rng(0)
for nx, ny in sizes:
    st8 = [rng.standard_normal((ny, nx)) * 0.3 for _ in range(6)]
    c = [rng.standard_normal((ny, nx)) * 5 for _ in range(3)]
    line = "%4dx%-4d" % (nx, ny)
    for rows in (1, 2, 4, 8, 16, 32, 64):
        ctx.set_option("rows_per_wave", rows)
        best = 1e9
        for rep in range(2):
            a = [x.copy() for x in st8]
            ctx.tvl1_iterations(*a, *c, 0.25, 0.15, 0.3, 100)
            s = ctx.stats()
            best = min(best, s.iter_ms[0] * 1e3 / s.iter_launches[0])
        line += "  r%-2d %7.2fus" % (rows, best)
    bpp = 120.0 if prec == ofx.F64 else 60.0
    print(line, flush=True)
