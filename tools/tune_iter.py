#!/usr/bin/env python3
"""Average time per TV-L1 inner iteration (HIP events over 100 fixed iterations) by image size:
one-iteration kernel vs the fused two-iteration kernel, swept over the strip height."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
prec = ofx.F32 if "--f32" in sys.argv else ofx.F64
sizes = [(3840, 2160), (1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]
quick = "--quick" in sys.argv
if quick:
    sizes = [(3840, 2160), (1920, 1080), (480, 270)]
ctx = ofx.Ofx(0, prec)
ctx.set_option("profile", 1)
rng = np.random.default_rng(0)
bpp = 120.0 if prec == ofx.F64 else 60.0


def run(nx, ny, st8, c):
    best = 1e9
    for rep in range(2):
        a = [x.copy() for x in st8]
        ctx.tvl1_iterations(*a, *c, 0.25, 0.15, 0.3, 100)
        s = ctx.stats()
        best = min(best, s.iter_ms[0] * 1e3 / s.iter_launches[0])
    return best


for nx, ny in sizes:
    st8 = [rng.standard_normal((ny, nx)) * 0.3 for _ in range(6)]
    c = [rng.standard_normal((ny, nx)) * 5 for _ in range(3)]
    ctx.set_option("fuse2", 0)
    ctx.set_option("rows_per_wave", 0)
    t1 = run(nx, ny, st8, c)
    line = "%4dx%-4d single(auto) %7.2fus %5.2f TB/s |" % (nx, ny, t1, bpp * nx * ny / t1 / 1e6)
    ctx.set_option("fuse2", 1)
    for rows in ((0, 8, 16) if quick else (0, 2, 4, 8, 16, 32)):
        ctx.set_option("rows_per_wave2", rows)
        t = run(nx, ny, st8, c)
        line += " r%-2d %7.2f" % (rows, t)
    ctx.set_option("rows_per_wave2", 0)
    print(line, flush=True)
