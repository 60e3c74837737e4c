#!/bin/bash
# round 3 session 11: interior fast path of the windowed SOR kernels: SOR suite + cfg 3 / cfg 4 lone and grouped
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sor.py tests/test_gpu_golden_cli.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/suite_sor.log 2>&1; echo "sor suite rc=$?"; tail -3 $O/suite_sor.log
for cfg in hs_cfg3 brox_cfg4; do
  echo "== $cfg"
  timeout -k 10 300 python tools/bench_sor_groups.py --only=$cfg --grid=1x1,1x16,2x16 --check 2>&1 | grep -v amdgpu.ids | cut -c1-230
done > $O/sor_fast.txt 2>&1; cat $O/sor_fast.txt
timeout -k 10 200 python tools/bench_sor.py --no-per-step 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -3
