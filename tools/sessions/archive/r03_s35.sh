#!/bin/bash
# round 3 session 35: the driver's 20 pairs over 2..6 streams (contexts) with the three-iteration kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ai; mkdir -p $O
cd $R
for st in 4 2 3 5 6 4; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $st --no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single --fixed-steps 0 > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('streams', $st, 'value', d['value'], d['repetitions']['seconds'], d['config'].get('lockstep_group'), d['config'].get('pairs_in_flight_per_gpu'))"
done | tee $O/streams.txt
