#!/usr/bin/env python3
"""tools/relaxed_accuracy.py: the f64 "tolerance" mode (option relaxed_dual = 1: double storage, sqrt(x^2 + y^2) instead of the
glibc-exact hypot and one reciprocal per denominator in the dual update) against the strict mode, which is bit-identical to the
oracle (tests/test_gpu_tvl1.py).  Prints one JSON line per configuration: AEPE, max |delta|, iteration-table agreement."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
PAR = dict(tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5, epsilon=0.01)
ctx = ofx.Ofx(0, ofx.F64)
cases = [("cfg1 640x480 P0", "P0", 640, 480, 0), ("cfg1 640x480 P1", "P1", 640, 480, 0), ("cfg2 1920x1080 P0", "P0", 1920, 1080, 0)]
cases += [("cfg2 1920x1080 P1 variant %d" % k, "P1", 1920, 1080, k) for k in range(8)]
cases += [("cfg5 3840x2160 P1 variant 1", "P1", 3840, 2160, 1)]
worst = 0.0
for name, pair, nx, ny, k in cases:
    if nx >= 3840:
        d0, d1 = synth.pair_device(pair, nx, ny, k, torch.device("cuda"))
    else:
        I0, I1 = synth.pair(pair, nx, ny, k)
        d0, d1 = torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda()
    flo = torch.zeros((2, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    its = []
    for m in (0, 1):
        ctx.set_option("relaxed_dual", m)
        ctx.tvl1_multiscale_dev(d0.data_ptr(), d1.data_ptr(), flo[m].data_ptr(), nx, ny, **PAR)
        ctx.synchronize()
        its.append(ctx.stats().iterations().copy())
    ctx.set_option("relaxed_dual", 0)
    a, b = flo[0].double(), flo[1].double()
    aepe = float(torch.sqrt(((a - b) ** 2).sum(-1)).mean())
    worst = max(worst, aepe)
    print(json.dumps({"case": name, "aepe_vs_strict": aepe, "max_abs": float((a - b).abs().max()),
                      "iteration_tables_equal": bool(np.array_equal(its[0], its[1])),
                      "loops_with_different_count": int((its[0] != its[1]).sum()), "loops": int(its[0].size),
                      "iterations_strict": int(its[0].sum()), "iterations_relaxed": int(its[1].sum())}))
print(json.dumps({"worst_aepe": worst, "tolerance": 1e-4, "holds": worst < 1e-4}))
