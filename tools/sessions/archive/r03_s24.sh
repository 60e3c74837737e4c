#!/bin/bash
# round 3 session 24: three iterations per launch (k_tvl1_iter3, option fuse3): parity through the TV-L1 suite, then A/B of the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03x; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q -k "fuse3 or strips" > $O/tvl1_tests.log 2>&1; rc=$?; echo "tvl1 tests rc=$rc"; tail -8 $O/tvl1_tests.log
[ $rc -ne 0 ] && exit 1
for v in "" "--opt fuse3=1" "" "--opt fuse3=1"; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ $v > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('opt', '$v', 'value', d['value'], 'other', (d.get('strict') or {}).get('value'), 'single', d['single_pair']['device_resident'], 'fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'roof4k', d.get('roofline_4k',{}).get('frac'))"
done | tee $O/ab.txt
