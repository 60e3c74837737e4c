#!/bin/bash
# HBM-side traffic of the SOR sweep kernels re-collected on the final sources (k_hs_tile with two barriers per sweep): r04_06 + r04_13, merged.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/r04_sor_traffic gpurun_out/r04_sor_traffic_rb
GRAFT_REPO_ROOT=$R bash tools/sessions/r04_06_sor_traffic.sh > gpurun_out/r04_sor_traffic_final.log 2>&1
GRAFT_REPO_ROOT=$R bash tools/sessions/r04_13_brox_rb_traffic.sh >> gpurun_out/r04_sor_traffic_final.log 2>&1
python3 - <<PY
import json
a = json.load(open("gpurun_out/r04_sor_traffic/summary.json"))
b = json.load(open("gpurun_out/r04_sor_traffic_rb/summary.json"))
assert a["kernel_source_sha16"] == b["kernel_source_sha16"]
for k, v in b.items():
    if k not in ("note", "kernel_source_sha16"):
        a[k] = v
json.dump(a, open("gpurun_out/r04_sor_traffic_merged.json", "w"), indent=1)
print(a["kernel_source_sha16"], sorted(k for k in a if k not in ("note", "kernel_source_sha16")))
PY
tail -14 gpurun_out/r04_sor_traffic_final.log | cut -c1-200
