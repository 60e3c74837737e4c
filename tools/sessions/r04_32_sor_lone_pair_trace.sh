#!/bin/bash
# Where a lone pair's time goes in the SOR tolerance mode: kernel traces of tools/sor_one_pair.py (6 solves), busy fraction and per-kernel time.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_sor_lone
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in hs brox; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$w -- python3 $R/tools/sor_one_pair.py $w --exact=0 --reps=5 > $OUT/$w.log 2>&1
  python3 $R/tools/trace_busy.py $OUT/$w 0.2 > $OUT/${w}_busy.txt 2>&1
  tail -1 $OUT/$w.log | cut -c1-300
  head -30 $OUT/${w}_busy.txt | cut -c1-160
done
find $OUT -name '*_kernel_trace.csv' -delete
