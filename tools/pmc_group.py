#!/usr/bin/env python3
"""Driver for PMC passes on the iteration kernel AS THE JOB LAUNCHES IT: one fixed-work solve (1 scale, 1 warp = 150 launches of
k_tvl1_iter2, two iterations each -- or 100 of k_tvl1_iter3, which the library picks for lockstep groups in tolerance mode unless
fuse3=0) of a lockstep group of G pairs at one size.   usage: pmc_group.py 1920x1080 G=16 [--f32] [relaxed=1] [fuse3=0|1|2]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
prec, tdt = (ofx.F32, torch.float32) if "--f32" in sys.argv else (ofx.F64, torch.float64)
nx, ny, G = 1920, 1080, 16
for a in sys.argv[1:]:
    if "x" in a and a[0].isdigit():
        nx, ny = map(int, a.split("x"))
    if a.startswith("G="):
        G = int(a[2:])
    if a.startswith("nt="):
        nt = int(a[3:])
    if a.startswith("relaxed="):
        relaxed = int(a[8:])
    if a.startswith("fuse3="):
        fuse3 = int(a[6:])
dev = torch.device("cuda:0")
ctx = ofx.Ofx(0, prec)
I0, I1, out = [], [], []
for k in range(G):
    a, b = synth.pair_device("P1", nx, ny, k, dev, tdt)
    I0.append(a); I1.append(b); out.append(torch.empty((ny, nx, 2), dtype=torch.float32, device=dev))
ctx.set_option("concurrency", 1)
if "nt" in dir():
    ctx.set_option("nt_stores", nt)
if "relaxed" in dir():
    ctx.set_option("relaxed_dual", relaxed)
ctx.set_option("fuse3", fuse3 if "fuse3" in dir() else 0)
ctx.set_option("fixed_work", 1)
ctx.tvl1_group_dev([t.data_ptr() for t in I0], [t.data_ptr() for t in I1], [t.data_ptr() for t in out], nx, ny, nscales=1, warps=1)
ctx.synchronize()
print("done", nx, ny, G)
