#!/bin/bash
# The driver's command with all its legs but the cli one (which starts child processes) under rocprofv3 --kernel-trace --stats: per-kernel totals of the whole run.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_full_stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cli > $OUT/bench.json 2> $OUT/bench.err
cd $R
python3 tools/fmt_kernel_stats.py $OUT/trace > $OUT/kernel_stats.txt 2>/dev/null
find $OUT -name '*_kernel_trace.csv' -delete
head -30 $OUT/kernel_stats.txt | cut -c1-170
