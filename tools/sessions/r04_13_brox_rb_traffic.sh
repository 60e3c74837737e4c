#!/bin/bash
# HBM-side traffic of Brox's red-black levels: k_brox_sor (two launches per sweep) against k_brox_tile (K = 4 / 1 sweeps per launch)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_sor_traffic_rb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BROX_TOL=1e-4
run() {
  tag=$1; shift
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/pmc_sor.py "$@" > $OUT/fetch_$tag.log 2>&1 || { echo "pass fetch_$tag rc=$?"; return 1; }
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$tag -- python3 $R/tools/pmc_sor.py "$@" > $OUT/write_$tag.log 2>&1 || { echo "pass write_$tag rc=$?"; return 1; }
  echo "done $tag: $(grep pixel_sweeps $OUT/fetch_$tag.log | cut -c1-200)"
}
run brox_redblack_two_launches_per_sweep brox G=16 sor_exact=0 sor_wave_levels=0 sor_fuse=9 &&
run brox_redblack_tile_k4 brox G=16 sor_exact=0 sor_wave_levels=0 sor_fuse=4 &&
run brox_redblack_tile_k1 brox G=16 sor_exact=0 sor_wave_levels=0 sor_fuse=1
cd $R
python3 tools/pmc_sor_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err
find $OUT -name '*.csv' -delete
python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in d.items():
    if k == "note": continue
    for n, r in v["kernels"].items():
        print("%-40s %-16s dispatches %5d  %.1f B per pixel-sweep = %.2f x compulsory" % (k, n, r["dispatches"], r["bytes_per_pixel_sweep"], r["traffic_over_compulsory"]))
PY
