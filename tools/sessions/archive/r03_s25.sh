#!/bin/bash
# round 3 session 25: fuse3 -- which levels (fuse3_min_px), strip height; the driver's command without the side legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03y; mkdir -p $O
cd $R
run() {
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ --no-4k --no-other-mode "$@" > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 -c "
import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('opts', sys.argv[1:], 'value', d['value'], 'reps', d['repetitions']['seconds'] if isinstance(d.get('repetitions'), dict) else d.get('repetitions'), 'single_ms', d['single_pair']['device_resident']['ms_per_pair'], 'fixed', d['fixed_work']['value'])" "$@"
}
run
run --opt fuse3=1
run --opt fuse3=1 --opt fuse3_min_px=500000
run --opt fuse3=1 --opt fuse3_min_px=2000000
run --opt fuse3=1 --opt fuse3_min_px=5000000
run --opt fuse3=1 --opt fuse3_min_px=2000000 --opt rows_per_wave3=16
run --opt fuse3=1 --opt fuse3_min_px=2000000 --opt rows_per_wave3=32
run
