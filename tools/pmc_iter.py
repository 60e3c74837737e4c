#!/usr/bin/env python3
"""Tiny driver for PMC passes: N launches of k_tvl1_iter at one size (default 4K f64, 20 iterations)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
prec = ofx.F32 if "--f32" in sys.argv else ofx.F64
nx, ny, n = 3840, 2160, 20
for a in sys.argv[1:]:
    if "x" in a and a[0].isdigit():
        nx, ny = map(int, a.split("x"))
    if a.startswith("n="):
        n = int(a[2:])
    if a.startswith("rows="):
        rows = int(a[5:])
ctx = ofx.Ofx(0, prec)
if "rows" in dir():
    ctx.set_option("rows_per_wave", rows)
rng = np.random.default_rng(0)
st8 = [rng.standard_normal((ny, nx)) * 0.3 for _ in range(6)]
c = [rng.standard_normal((ny, nx)) * 5 for _ in range(3)]
ctx.tvl1_iterations(*st8, *c, 0.25, 0.15, 0.3, n)
print("done", nx, ny, n)
