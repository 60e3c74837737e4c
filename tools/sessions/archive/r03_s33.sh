#!/bin/bash
# round 3 session 33: contexts alternating between the two window kernels -- three contexts, both solvers
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ag; mkdir -p $O
cd $R
for spec in "--only=hs_cfg3 --grid=3x16" "--only=hs_cfg3 --grid=3x11" "--only=hs_cfg3 --grid=3x11 --opt-odd=sor_lds=2 --opt-odd=sor_rows=61" "--only=hs_cfg3 --grid=3x11 --opt-odd=sor_lds=2" "--only=hs_cfg3 --grid=3x16 --opt-odd=sor_lds=2" \
            "--only=brox_cfg4 --grid=2x16" "--only=brox_cfg4 --grid=3x16" "--only=brox_cfg4 --grid=3x16 --opt-odd=sor_lds=2" "--only=brox_cfg4 --grid=3x11 --opt-odd=sor_lds=2"; do
  echo "== $spec"
  timeout -k 10 300 python tools/bench_sor_groups.py $spec 2>&1 | grep config | cut -c1-260 || exit 1
done | tee $O/mixed.txt
