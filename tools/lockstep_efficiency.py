#!/usr/bin/env python3
"""How much of a lockstep group's iteration time is real work?  A group iterates a (level, warp) loop until its LAST pair
has stopped; pairs that stopped earlier ride along masked.  efficiency = sum over pairs of their own iterations x pixels /
(G x sum over loops of the group's longest loop x pixels), for the bench job's pairs (1080p P1 batch variants)."""
import importlib, json, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
nx, ny, ns, warps = 1920, 1080, 5, 5
ctx = ofx.Ofx(0, ofx.F64)
dev = torch.device("cuda:0")
for G in (4, 5, 8, 16):
    I0, I1, out = [], [], []
    for k in range(G):
        a, b = synth.pair_device("P1", nx, ny, k, dev, torch.float64)
        I0.append(a); I1.append(b); out.append(torch.empty((ny, nx, 2), dtype=torch.float32, device=dev))
    st = ctx.tvl1_group_dev([t.data_ptr() for t in I0], [t.data_ptr() for t in I1], [t.data_ptr() for t in out], nx, ny)
    it = np.array([[[s.iters[l][w] for w in range(warps)] for l in range(ns)] for s in st])          # [pair][level][warp]
    px = np.array([st[0].nx[l] * st[0].ny[l] for l in range(ns)], dtype=np.float64)
    own = (it * px[None, :, None]).sum()
    lock = (it.max(axis=0) * px[:, None]).sum() * G
    print(json.dumps({"group": G, "efficiency": round(float(own / lock), 4), "finest_level_iterations_per_pair": it[:, 0, :].tolist()[:4],
                      "finest_max": it[:, 0, :].max(axis=0).tolist(), "finest_mean": np.round(it[:, 0, :].mean(axis=0), 1).tolist()}), flush=True)
