#!/bin/bash
# round 3 session 49: iterations per launch of the fused chi solver (halo = iterations): 4 / 5 / 6 / 7
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03aw; mkdir -p $O
cd $R
for rep in 1 2; do for v in "" chi4 chi6 chi7; do
  lib=""; [ -n "$v" ] && lib="OFX_LIB_PATH=$R/variants/libofx_$v.so"
  env $lib timeout -k 10 300 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --cpu none 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
print('variant', '${v:-chi5}', [(json.loads(l)['size'], json.loads(l)['gpu_s']) for l in sys.stdin])" || exit 1
done; done | tee $O/chi_n.txt
