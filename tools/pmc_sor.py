#!/usr/bin/env python3
"""ONE full-resolution single-scale solve of a lockstep group with a fixed number of sweeps, for rocprofv3 --pmc passes on the SOR
kernels (a whole multiscale solve is thousands of dispatches and a counter pass costs > 60 ms per dispatch).
Usage: pmc_sor.py hs|brox G=16 sweeps=4 [name=value options ...]   (e.g. sor_exact=1 sor_lds=0 | sor_exact=0 sor_fuse=2)
Horn-Schunck: 1920x1080, one warp; Brox: 1280x720, one outer / inner iteration.  TOL = 0: every pair runs all `sweeps` sweeps, so
the pixel-sweeps of the run are G x nx x ny x sweeps (printed, and read by tools/pmc_sor_summary.py from the .log)."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
which = sys.argv[1]
G, sweeps, opts = 16, 4, {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    if k == "G": G = int(v)
    elif k == "sweeps": sweeps = int(v)
    else: opts[k] = float(v)
dev = torch.device("cuda", 0)
nx, ny = (1920, 1080) if which == "hs" else (1280, 720)
ins = [synth.pair_device("P0" if k == 0 else "P1", nx, ny, k, dev) for k in range(G)]
flo = torch.empty((G, ny, nx, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = ofx.Ofx(0, ofx.F64)
for k, v in opts.items():
    ctx.set_option(k, v)
args = ([t[0].data_ptr() for t in ins], [t[1].data_ptr() for t in ins], [flo[k].data_ptr() for k in range(G)], nx, ny)
if which == "hs":
    st = ctx.hs_group_dev(*args, alpha=20.0, nscales=1, zfactor=0.5, warps=1, TOL=0.0, maxiter=sweeps)
else:
    # the Brox entry has no maxiter argument (the reference's 300 is a constant): a tolerance the first sweeps reach is not
    # available either, so the solve runs `sweeps` = what TOL lets it run; TOL is chosen large enough to stop early
    st = ctx.brox_group_dev(*args, alpha=50.0, gamma=10.0, nscales=1, nu=0.5, TOL=float(os.environ.get("BROX_TOL", "2e-3")), inner=1, outer=1)
ctx.synchronize()
n = [int(s.iterations().sum()) for s in st]
print(json.dumps({"which": which, "G": G, "nx": nx, "ny": ny, "sweeps_per_pair": n, "pixel_sweeps": float(sum(n)) * nx * ny, "options": opts}))
