#!/bin/bash
# session 24: group-launch timing after the store_a guard (fixed-work launches no longer store the intermediate state at e1 == 0)
set -o pipefail
timeout -k 10 300 python tools/group_roofline.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_gpu_tvl1.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-sor 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'fixed', d['fixed_work']['value']); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','avg_launch_us','pairs_per_launch')}, r['single_pair']); r=d['roofline_4k']; print({k:r[k] for k in ('achieved','frac','avg_launch_us','pairs_per_launch')}, r['single_pair'])"
