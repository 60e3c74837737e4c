#!/bin/bash
# session 12: TV-L1 with occlusions, whole solve on the device: parity + first timing
set -o pipefail
mkdir -p gpurun_out/r02l
timeout -k 10 500 python -m pytest tests/test_gpu_occ.py -x -q > gpurun_out/r02l/occ_tests.log 2>&1
echo "occ tests: $?"; tail -3 gpurun_out/r02l/occ_tests.log
timeout -k 10 400 python tools/bench_tvl1occ.py --size 160x120 --size 320x240 --size 640x480 --check > gpurun_out/r02l/tvl1occ.jsonl 2> gpurun_out/r02l/tvl1occ.err
echo "bench: $?"; cat gpurun_out/r02l/tvl1occ.jsonl; tail -3 gpurun_out/r02l/tvl1occ.err
