#!/bin/bash
# round 3 session 5: suite (tile off by default, both SOR window kernels forced, staged batch inputs), HS / Brox windows in LDS
# against the global kernels, lone solves and groups
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -5 $O/suite.log
for cfg in hs_cfg3 brox_cfg4; do
for v in "sor_lds=0" "sor_lds=2 --opt=sor_window=8" "sor_lds=2 --opt=sor_window=16" "sor_lds=2 --opt=sor_window=8 --opt=sor_rows=64"; do
  echo "== $cfg $v"
  timeout -k 10 300 python tools/bench_sor_groups.py --only=$cfg --grid=1x1,1x16,2x16 --opt=$v 2>&1 | grep -v amdgpu.ids | cut -c1-230
done; done > $O/sor_lds.txt 2>&1; cat $O/sor_lds.txt
