#!/bin/bash
# session 19: bench with the roofline quoted on group launches + bench tests
set -o pipefail
mkdir -p gpurun_out/r02s
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r02s/bench20.json 2> gpurun_out/r02s/bench20.err || { tail -5 gpurun_out/r02s/bench20.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02s/bench20.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps")})
for k in ("roofline", "roofline_4k"):
    r = d[k]; print(k, {x: r[x] for x in ("kernel", "achieved", "frac", "avg_launch_us", "launches", "pairs_per_launch", "traffic", "hbm_frac_counter") if x in r}, r.get("single_pair"))
PY
timeout -k 10 800 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3
