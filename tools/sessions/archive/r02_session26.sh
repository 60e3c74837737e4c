#!/bin/bash
# session 26: the driver's 20-step command under other stream counts (groups = ceil(20 / streams) pairs)
for st in 4 2 3 5 4; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --streams $st --no-cpu --no-4k --no-sor --no-occ --fixed-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', d['config']['streams_per_gpu'], 'lockstep', d['config']['lockstep_group'], 'value', d['value'], 'ms_per_step', d['ms_per_step'])"
done
