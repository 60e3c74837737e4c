#!/usr/bin/env python3
"""Host-buffer entry point vs device-resident entry point on the bench workload: what PCIe costs."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
nx, ny = 1920, 1080
I0, I1 = synth.pair("P1", nx, ny)
c = ofx.Ofx(0, ofx.F64)
d0, d1 = torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda()
flo = torch.empty((ny, nx, 2), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    c.tvl1_multiscale(I0, I1)
    c.tvl1_multiscale_dev(d0.data_ptr(), d1.data_ptr(), flo.data_ptr(), nx, ny)
c.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    c.tvl1_multiscale(I0, I1)
th = (time.perf_counter() - t0) / 10
work = c.stats().work_pix_iters
t0 = time.perf_counter()
for _ in range(10):
    c.tvl1_multiscale_dev(d0.data_ptr(), d1.data_ptr(), flo.data_ptr(), nx, ny)
c.synchronize()
td = (time.perf_counter() - t0) / 10
print("host-pointer API %.2f ms/pair = %.0f Mpix*warp-iters/s ; device-resident %.2f ms/pair = %.0f ; PCIe + staging %.2f ms"
      % (th * 1e3, work / th / 1e6, td * 1e3, work / td / 1e6, (th - td) * 1e3))
