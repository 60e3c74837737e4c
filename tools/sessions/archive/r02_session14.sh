#!/bin/bash
# session 14: ROF kernel iterations: occ parity tests, tvl1occ timing, kernel stats
set -o pipefail
mkdir -p gpurun_out/r02n
timeout -k 10 400 python -m pytest tests/test_gpu_occ.py -x -q > gpurun_out/r02n/occ_tests.log 2>&1
rc=$?; echo "occ tests: $rc"; tail -3 gpurun_out/r02n/occ_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_tvl1occ.py --size 320x240 --size 640x480 --size 1920x1080 --cpu none > gpurun_out/r02n/tvl1occ.jsonl 2> gpurun_out/r02n/tvl1occ.err || exit 1
cat gpurun_out/r02n/tvl1occ.jsonl
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/r02n/occ
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02n/occ -- python3 tools/bench_tvl1occ.py --size 640x480 --cpu none > gpurun_out/r02n/occ.jsonl 2> gpurun_out/r02n/occ.err
find gpurun_out/r02n -name "*kernel_trace.csv" -delete
find gpurun_out/r02n -name "*kernel_stats.csv" | head -1 | xargs -r head -6
