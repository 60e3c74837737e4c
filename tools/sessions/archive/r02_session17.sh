#!/bin/bash
# session 17: numbers for DESIGN 5.6: tvl1occ against the reference (16 cores and 1 core) at three sizes, Solver_wrt_u alone
set -o pipefail
mkdir -p gpurun_out/r02q
rm -f gpurun_out/r02q/*.jsonl
timeout -k 10 600 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --cpu ref --check > gpurun_out/r02q/tvl1occ_vs_ref.jsonl 2> gpurun_out/r02q/err.txt || { tail -3 gpurun_out/r02q/err.txt; exit 1; }
cat gpurun_out/r02q/tvl1occ_vs_ref.jsonl
timeout -k 10 300 python - <<'PY' | tee gpurun_out/r02q/solver_u.jsonl
import importlib, json, sys, time
import numpy as np
sys.path.insert(0, ".")
ofx = importlib.import_module("optical-flow-1_amd")
import oracle
ctx = ofx.Ofx(0, ofx.F64)
ref = oracle.Ref()
for nx, ny in ((640, 480), (1920, 1080)):
    rng = np.random.default_rng(1)
    v1, v2 = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    chi = np.clip(rng.random((ny, nx)) * 1.4 - 0.2, 0, 1)
    g = 1.0 / (1.0 + rng.random((ny, nx)) * 3)
    ctx.occ_solver_u(v1, v2, chi, g, 0.3, 0.15, n_iter=1)
    t = time.perf_counter(); r = ctx.occ_solver_u(v1, v2, chi, g, 0.3, 0.15, n_iter=10); gs = time.perf_counter() - t
    ref.set_num_threads(1)
    t = time.perf_counter(); o = ref.occ_solver_u(v1, v2, chi, g, 0.3, 0.15, fresh=True); cs = time.perf_counter() - t
    print(json.dumps({"solver_wrt_u": "%dx%d" % (nx, ny), "gpu_s_incl_transfers": round(gs, 4), "reference_1_thread_s": round(cs, 3),
                      "identical": bool(np.array_equal(r[0], o[0]) and np.array_equal(r[1], o[1]))}), flush=True)
PY
