#!/bin/bash
# session 16: tvl1occ batch over contexts: parity + throughput
set -o pipefail
mkdir -p gpurun_out/r02p
timeout -k 10 400 python -m pytest tests/test_gpu_occ.py -x -q > gpurun_out/r02p/occ_tests.log 2>&1
rc=$?; echo "occ tests: $rc"; tail -3 gpurun_out/r02p/occ_tests.log
[ $rc -eq 0 ] || exit 1
for b in 1:16 2:32 4:64; do
  timeout -k 10 300 python tools/bench_tvl1occ.py --size 640x480 --cpu none --batch $b >> gpurun_out/r02p/batch.jsonl 2>> gpurun_out/r02p/batch.err || exit 1
done
timeout -k 10 300 python tools/bench_tvl1occ.py --size 1920x1080 --cpu none --batch 1:14 >> gpurun_out/r02p/batch.jsonl 2>> gpurun_out/r02p/batch.err || exit 1
cat gpurun_out/r02p/batch.jsonl
