#!/bin/bash
# session 29: the fused kernel compiled for 4 waves per SIMD (128 VGPRs, 24 dwords spilled) against the production 3 (154 VGPRs)
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" $R/variants/libofx_w4.so; do
  [ -z "$lib" ] && unset OFX_LIB_PATH || export OFX_LIB_PATH=$lib
  echo "== ${lib:-production}"
  timeout -k 10 200 python tools/group_rows_sweep.py 1920x1080 G=16 2>&1 | grep -v amdgpu.ids | head -1
  timeout -k 10 200 python tools/group_rows_sweep.py 3840x2160 G=4 2>&1 | grep -v amdgpu.ids | head -1
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-4k --no-sor --no-occ --fixed-steps 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value', d['value'], 'fixed', d['fixed_work']['value'], 'single-pair launch us', d['roofline']['single_pair']['avg_launch_us'])"
done
