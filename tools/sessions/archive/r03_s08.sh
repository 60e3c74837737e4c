#!/bin/bash
# round 3 session 8: strip height of the fused kernel in the (memory-bound) tolerance mode, A/B on one box; host-entry figure with
# reused result planes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout -k 10 900 python tools/ab_bench.py auto= r28=,rows_per_wave2=28 r40=,rows_per_wave2=40 r54=,rows_per_wave2=54 --rounds 2 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single" > $O/ab_rows.txt 2>&1; cat $O/ab_rows.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ --no-4k > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'strict', d['strict']['value']); print('single', d['single_pair'])"
