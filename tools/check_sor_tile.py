#!/usr/bin/env python3
"""Development check of the tolerance-mode tile sweeps (ofx_sor_tile.hip): bit-compare with the oracle in the same sweep order on
small inputs, then time BASELINE config 3 / 4 for every K.  Usage: check_sor_tile.py [--quick] [--no-big]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
import oracle

ctx = ofx.Ofx(0, ofx.F64)
orc = oracle.Oracle()
orc.set_num_threads(1)
bad = 0


def aepe(a, b):
    return float(np.mean(np.hypot(a[0] - b[0], a[1] - b[1])))


ctx.set_option("sor_exact", 0)
orc.set_sor_order(1)
for (pair, nx, ny, ns) in (("P0", 64, 48, 2), ("P1", 135, 68, 3), ("P1", 33, 47, 2), ("P1", 300, 130, 2), ("P0", 9, 8, 1),
                           ("P1", 257, 75, 1)):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=20.0, nscales=ns, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
    uo, vo, it_o = orc.hs_pyramidal(I1, I2, **kw)
    for K, geom in [(-1, 0)] + [(K, geom) for geom in (1, 2, 3) for K in (1, 2, 3, 4)]:
        ctx.set_option("sor_fuse", K)
        ctx.set_option("sor_tile", geom)
        ug, vg = ctx.hs_pyramidal(I1, I2, **kw)
        it = ctx.stats().iterations()
        ok = np.array_equal(it, it_o) and np.array_equal(ug, uo) and np.array_equal(vg, vo)
        bad += not ok
        print("hs %s %dx%d ns=%d K=%d geom=%d: %s  sweeps %d vs %d  max|d| %.3g" % (pair, nx, ny, ns, K, geom, "ok" if ok else "MISMATCH", int(it.sum()),
              int(np.asarray(it_o).sum()), max(np.abs(ug - uo).max(), np.abs(vg - vo).max())), flush=True)
        if not ok:
            print("   gpu", it.tolist(), "\n   orc", np.asarray(it_o).tolist(), flush=True)
ctx.set_option("sor_fuse", 0)
ctx.set_option("sor_tile", 0)
# Brox: checkerboard of tiles on the finest level(s), red-black below
orc.set_sor_wave_levels(1)
for (pair, nx, ny, ns, tw, wl) in (("P0", 64, 48, 2, 64, 1), ("P1", 160, 120, 3, 64, 1), ("P1", 135, 68, 2, 32, 2), ("P1", 300, 130, 2, 128, 1), ("P0", 33, 70, 1, 16, 1),
                                  ("P1", 257, 75, 2, 64, 2)):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=ns, nu=0.5, TOL=1e-4, inner=2, outer=4)
    orc.set_sor_tile(tw, 64)
    orc.set_sor_wave_levels(wl)
    ctx.set_option("sor_tile_w", tw)
    ctx.set_option("sor_wave_levels", wl)
    uo, vo, it_o = orc.brox_spatial(I1, I2, **kw)
    ug, vg = ctx.brox_spatial(I1, I2, **kw)
    it = ctx.stats().iterations()
    ok = np.array_equal(it, it_o) and np.array_equal(ug, uo) and np.array_equal(vg, vo)
    bad += not ok
    print("brox %s %dx%d ns=%d tw=%d wave_levels=%d: %s  sweeps %d vs %d  max|d| %.3g" % (pair, nx, ny, ns, tw, wl, "ok" if ok else "MISMATCH", int(it.sum()),
          int(np.asarray(it_o).sum()), max(np.abs(ug - uo).max(), np.abs(vg - vo).max())), flush=True)
    if not ok:
        print("   gpu", it.tolist(), "\n   orc", np.asarray(it_o).tolist(), flush=True)
ctx.set_option("sor_tile_w", 0)
ctx.set_option("sor_wave_levels", 1)
orc.set_sor_tile(64, 64)
orc.set_sor_wave_levels(1)
if "--no-big" not in sys.argv:
    nx, ny = 1920, 1080
    I1, I2 = synth.pair("P0", nx, ny)
    kw = dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    orc.set_sor_order(0)
    t0 = time.perf_counter()
    ref = orc.hs_pyramidal(I1, I2, **kw)
    print("oracle (reference order, 1 thread) %.2f s, sweeps %d" % (time.perf_counter() - t0, int(np.asarray(ref[2]).sum())), flush=True)
    for K, geom in [(-1, 0)] + [(K, geom) for geom in (1, 2, 3) for K in (1, 2, 3, 4)]:
        ctx.set_option("sor_fuse", K)
        ctx.set_option("sor_tile", geom)
        ctx.hs_pyramidal(I1, I2, **kw)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ug, vg = ctx.hs_pyramidal(I1, I2, **kw)
            ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        print(json.dumps({"cfg3": "hs 1920x1080", "K": K, "geom": geom, "seconds": round(min(ts), 4), "sweeps": int(st.iterations().sum()),
                          "per_level": [int(x) for x in st.iterations().sum(axis=1)],
                          "mpix_sweeps_per_s": round(st.work_pix_iters / min(ts) / 1e6, 1),
                          "aepe_vs_reference_order": aepe((ug, vg), ref)}), flush=True)
if "--no-big" not in sys.argv:
    nx, ny = 1280, 720
    I1, I2 = synth.pair("P0", nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)
    orc.set_sor_order(0)
    t0 = time.perf_counter()
    ref = orc.brox_spatial(I1, I2, **kw)
    print("oracle brox (reference order, 1 thread) %.2f s, sweeps %d" % (time.perf_counter() - t0, int(np.asarray(ref[2]).sum())), flush=True)
    for wl, tw, K in ((0, 128, 9), (0, 128, 2), (1, 128, 9), (1, 128, 1), (1, 128, 2), (1, 128, 4), (1, 64, 2)):
        ctx.set_option("sor_wave_levels", wl)
        ctx.set_option("sor_tile_w", tw)
        ctx.set_option("sor_fuse", K)
        ctx.brox_spatial(I1, I2, **kw)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ug, vg = ctx.brox_spatial(I1, I2, **kw)
            ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        print(json.dumps({"cfg4": "brox 1280x720", "wave_levels": wl, "tile_w": tw, "K": K, "seconds": round(min(ts), 4), "sweeps": int(st.iterations().sum()),
                          "per_level": [int(x) for x in st.iterations().sum(axis=1)],
                          "mpix_sweeps_per_s": round(st.work_pix_iters / min(ts) / 1e6, 1),
                          "aepe_vs_reference_order": aepe((ug, vg), ref)}), flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
