#!/usr/bin/env python3
"""Strip height of the fused iteration kernel in GROUP launches (fixed work): us per launch and fraction of the HBM peak on the
compulsory bytes, for rows_per_wave2 = automatic (0), 16, 24, 32, 48, 64.   usage: group_rows_sweep.py 3840x2160 G=4"""
import importlib, json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
nx, ny, G = 3840, 2160, 4
for a in sys.argv[1:]:
    if "x" in a and a[0].isdigit():
        nx, ny = map(int, a.split("x"))
    if a.startswith("G="):
        G = int(a[2:])
dev = torch.device("cuda:0")
ctx = ofx.Ofx(0, ofx.F64)
I0, I1, out = [], [], []
for k in range(G):
    a, b = synth.pair_device("P1", nx, ny, k, dev, torch.float64)
    I0.append(a); I1.append(b); out.append(torch.empty((ny, nx, 2), dtype=torch.float32, device=dev))
args = ([t.data_ptr() for t in I0], [t.data_ptr() for t in I1], [t.data_ptr() for t in out], nx, ny)
ctx.set_option("concurrency", 1); ctx.set_option("profile", 1); ctx.set_option("fixed_work", 1)
for rows in (0, 16, 24, 32, 48, 64):
    ctx.set_option("rows_per_wave2", rows)
    ctx.tvl1_group_dev(*args, nscales=1, warps=1)
    st = ctx.tvl1_group_dev(*args, nscales=1, warps=2)
    us = st[0].iter_ms[0] * 1e3 / (st[0].iter_launches[0] / 2)
    print(json.dumps({"size": "%dx%d" % (nx, ny), "G": G, "rows_per_wave2": rows, "us_per_launch": round(us, 1),
                      "frac": round(120.0 * nx * ny * G / (us * 1e-6) / 8e12, 4)}), flush=True)
