#!/bin/bash
# HBM-side traffic of the SOR sweep kernels: one full-resolution single-scale group solve per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_sor_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # tag, then the arguments of tools/pmc_sor.py
  tag=$1; shift
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/pmc_sor.py "$@" > $OUT/fetch_$tag.log 2>&1 || { echo "pass fetch_$tag rc=$?"; return 1; }
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$tag -- python3 $R/tools/pmc_sor.py "$@" > $OUT/write_$tag.log 2>&1 || { echo "pass write_$tag rc=$?"; return 1; }
  echo "done $tag: $(grep pixel_sweeps $OUT/fetch_$tag.log | cut -c1-160)"
}
run hs_exact_global   hs G=16 sweeps=4 sor_exact=1 sor_lds=0 &&
run hs_exact_lds      hs G=16 sweeps=4 sor_exact=1 sor_lds=2 &&
run hs_tolerance_k2   hs G=16 sweeps=4 sor_exact=0 sor_fuse=2 &&
run hs_tolerance_k4   hs G=16 sweeps=4 sor_exact=0 sor_fuse=4 &&
run brox_exact_global brox G=16 sor_exact=1 sor_lds=0 &&
run brox_exact_lds    brox G=16 sor_exact=1 sor_lds=2 &&
run brox_tolerance    brox G=16 sor_exact=0 &&
run brox_redblack     brox G=16 sor_exact=0 sor_wave_levels=0
cd $R
python3 tools/pmc_sor_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_counter_collection.csv' -delete; find $OUT -name '*.csv' -size +1M -delete
python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in d.items():
    if k == "note": continue
    for n, r in v["kernels"].items():
        print("%-20s %-24s dispatches %5d  %.1f B per pixel-sweep = %.2f x compulsory" % (k, n, r["dispatches"], r["bytes_per_pixel_sweep"], r["traffic_over_compulsory"]))
PY
