#!/bin/bash
# round 3 session 34: SOR batches -- contexts x group size at 48 / 64 pairs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ah; mkdir -p $O
cd $R
for spec in "--only=hs_cfg3 --grid=4x12" "--only=hs_cfg3 --grid=6x8" "--only=hs_cfg3 --grid=4x16" "--only=hs_cfg3 --grid=3x16 --opt=sor_rows=64" "--only=hs_cfg3 --grid=3x16 --opt=sor_window=16" \
            "--only=brox_cfg4 --grid=4x12" "--only=brox_cfg4 --grid=4x16" "--only=brox_cfg4 --grid=6x8"; do
  echo "== $spec"
  timeout -k 10 300 python tools/bench_sor_groups.py $spec 2>&1 | grep config | cut -c1-260 || exit 1
done | tee $O/grid.txt
