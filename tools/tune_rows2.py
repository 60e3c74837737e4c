#!/usr/bin/env python3
"""Fine sweep of the fused kernel's strip height on real (P1) data at the level sizes of the 1080p / 4K pyramids."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
import oracle
orc = oracle.Oracle(); orc.set_num_threads(min(oracle.host_cores(), 16))
ctx = ofx.Ofx(0, ofx.F64)
ctx.set_option("profile", 1)
for (nx, ny), rows in (((3840, 2160), (8, 12, 16, 20, 24, 32)), ((1920, 1080), (6, 8, 9, 10, 11, 12, 13, 14, 16)),
                       ((960, 540), (2, 3, 4, 5, 6, 8)), ((480, 270), (1, 2, 3, 4)), ((240, 135), (1, 2, 3)), ((120, 68), (1, 2))):
    I0, I1 = synth.pair("P1", nx, ny)
    rng = np.random.default_rng(0)
    u1, u2 = rng.standard_normal((ny, nx)) * 0.5, rng.standard_normal((ny, nx)) * 0.5
    I1x, I1y = orc.centered_gradient(I1)
    I1w, I1wx, I1wy = (orc.bicubic_warp(x, u1, u2, True) for x in (I1, I1x, I1y))
    rho_c = I1w - I1wx * u1 - I1wy * u2 - I0
    p = [rng.standard_normal((ny, nx)) * 0.1 for _ in range(4)]
    line = "%4dx%-4d" % (nx, ny)
    for r in rows:
        ctx.set_option("rows_per_wave2", r)
        best = 1e9
        for rep in range(3):
            a = [x.copy() for x in (u1, u2, *p)]
            ctx.tvl1_iterations(*a, I1wx, I1wy, rho_c, 0.25, 0.15, 0.3, 100)
            s = ctx.stats()
            best = min(best, s.iter_ms[0] * 1e3 / s.iter_launches[0])
        line += "  r%-2d %6.2f" % (r, best)
    print(line, flush=True)
