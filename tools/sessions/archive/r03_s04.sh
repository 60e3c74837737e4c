#!/bin/bash
# round 3 session 4: tile kernel + LDS-staged HS windows (suite), A/B on one box: common-domain hypot on/off, tile kernel on/off;
# HS cfg 3 with sor_lds on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -5 $O/suite.log
for v in "sor_lds=0" "sor_lds=1 --opt=sor_window=8" "sor_lds=1 --opt=sor_window=16" "sor_lds=1 --opt=sor_window=24"; do
  echo "== hs cfg3 $v"
  timeout -k 10 300 python tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x1,1x16,2x16 --opt=$v 2>&1 | grep -v amdgpu.ids | cut -c1-230
done > $O/hs_lds.txt 2>&1; cat $O/hs_lds.txt
timeout -k 10 600 python tools/ab_bench.py prod= hypgen=variants/libofx_hypgen.so tile0=,tile=0 tile6=,tile=6 --rounds 2 > $O/ab.txt 2>&1; cat $O/ab.txt
