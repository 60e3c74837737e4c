#!/bin/bash
# round 3 session 32: do the two HS window kernels use complementary resources?  contexts alternating between them
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03af; mkdir -p $O
cd $R
for spec in "--grid=2x16" "--grid=2x16 --opt-odd=sor_lds=2 --opt-odd=sor_rows=61" "--grid=4x16 --opt-odd=sor_lds=2 --opt-odd=sor_rows=61" "--grid=3x16 --opt-odd=sor_lds=2 --opt-odd=sor_rows=61" "--grid=2x16 --opt=sor_lds=2 --opt=sor_rows=61"; do
  echo "== $spec"
  timeout -k 10 300 python tools/bench_sor_groups.py --only=hs_cfg3 $spec 2>&1 | grep config | cut -c1-260 || exit 1
done | tee $O/mixed.txt
