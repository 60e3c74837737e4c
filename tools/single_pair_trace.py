#!/usr/bin/env python3
"""tools/single_pair_trace.py [strict|tolerance] [opt=val ...]: N device-resident single-pair solves (1080p P1 variants, eps = 0.01) for a
rocprofv3 --kernel-trace run; tools/trace_budget.py <dir> --all then shows where one pair's time goes (kernels, gaps between launches)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
nx, ny = 1920, 1080
c = ofx.Ofx(0, ofx.F64)
c.set_option("relaxed_dual", 0 if "strict" in sys.argv else 1)
for a in sys.argv[1:]:
    if "=" in a:
        c.set_option(a.split("=")[0], float(a.split("=")[1]))
dev = torch.device("cuda")
pairs = [synth.pair_device("P1", nx, ny, k, dev) for k in range(4)]
flo = torch.empty((ny, nx, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for k in range(2):
    c.tvl1_multiscale_dev(pairs[k][0].data_ptr(), pairs[k][1].data_ptr(), flo.data_ptr(), nx, ny)
c.synchronize()
mark = torch.empty(7777, dtype=torch.float32, device=dev)
mark.fill_(1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 8
for k in range(N):
    c.tvl1_multiscale_dev(pairs[k % 4][0].data_ptr(), pairs[k % 4][1].data_ptr(), flo.data_ptr(), nx, ny)
    c.synchronize()
dt = (time.perf_counter() - t0) / N
mark.fill_(2.0)
torch.cuda.synchronize()
print("single pair %.3f ms" % (dt * 1e3), c.stats().iterations().tolist())
