#!/bin/bash
# session 21: what does cutting a rank's 20 pairs into two rounds (16 + 4, the gather of the first overlapping the second) cost on one GPU?
set -o pipefail
O=gpurun_out/r02u; mkdir -p $O
for r in 1 2 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --rounds $r --no-cpu --no-4k --no-sor --fixed-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rounds', d['config']['rounds_per_gpu'], 'value', d['value'], 'ms_per_step', d['ms_per_step'])"
done
