#!/bin/bash
# session 13: kernel stats of the TV-L1-with-occlusions solve at 640x480
set -o pipefail
mkdir -p gpurun_out/r02m
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02m/occ -- python3 tools/bench_tvl1occ.py --size 640x480 --cpu none > gpurun_out/r02m/occ.jsonl 2> gpurun_out/r02m/occ.err
echo "prof: $?"; cat gpurun_out/r02m/occ.jsonl
find gpurun_out/r02m -name "*kernel_trace.csv" -delete
find gpurun_out/r02m -name "*kernel_stats.csv" | head -1 | xargs -r head -14
