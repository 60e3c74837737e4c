#!/bin/bash
# round 3 session 46: lone Brox / HS solves with the two window kernels after the idle-unit exit
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03at; mkdir -p $O
cd $R
for spec in "--only=brox_cfg4 --opt=sor_lds=0" "--only=brox_cfg4 --opt=sor_lds=2" "--only=brox_cfg4 --opt=sor_lds=2 --opt=sor_window=4" "--only=hs_cfg3 --opt=sor_lds=0" "--only=hs_cfg3 --opt=sor_lds=2"; do
  echo "== $spec"
  timeout -k 10 300 python tools/bench_sor_groups.py --grid=1x1 $spec > $O/out.txt 2>&1; grep config $O/out.txt | cut -c1-240; tail -3 $O/out.txt | cut -c1-300
done | tee $O/lone.txt
