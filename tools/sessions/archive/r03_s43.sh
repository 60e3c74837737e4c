#!/bin/bash
# round 3 session 43: bench.py with no flags (64 steps, groups of 16) -- the other way the driver may call it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03aq; mkdir -p $O
cd $R
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
SECONDS=0; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'steps', d['steps'], d['config']['arithmetic_mode'], d['config'].get('lockstep_group'), d['repetitions']['seconds'], 'strict', d['strict']['value'], 'ms_per_step', d['ms_per_step'])
print('loop_ends', d.get('loop_ends'))
r=d['roofline']; print('roof', r['kernel'], r['frac'], r['avg_launch_us'])"
