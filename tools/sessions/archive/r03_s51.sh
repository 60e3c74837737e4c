#!/bin/bash
# round 3 session 51: Brox / HS batches on 3 contexts x 16 -- window length and rows per block
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ay; mkdir -p $O
cd $R
for spec in "--only=brox_cfg4" "--only=brox_cfg4 --opt=sor_window=8" "--only=brox_cfg4 --opt=sor_window=2" "--only=brox_cfg4 --opt=sor_rows=64" "--only=brox_cfg4 --opt=sor_rows=253" "--only=brox_cfg4 --opt=sor_window=8 --opt=sor_rows=253" \
            "--only=hs_cfg3 --opt=sor_window=4" "--only=hs_cfg3 --opt=sor_rows=253" "--only=hs_cfg3 --opt=sor_window=12"; do
  echo "== $spec"
  timeout -k 10 300 python tools/bench_sor_groups.py --grid=3x16 $spec > $O/out.txt 2>&1; grep config $O/out.txt | cut -c1-260; grep -i "error\|Traceback" $O/out.txt | head -2
done | tee $O/sweep.txt
