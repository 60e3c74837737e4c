#!/bin/bash
# round 3 session 42: final-tree record after the ROF window change -- full GPU suite, smoke, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ap; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/suite.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions']['seconds'], 'strict', d['strict']['value'])
print('single', d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'], 'fixed', d['fixed_work']['value'])
r=d['roofline']; print('roof', r['kernel'], r['frac'], r['avg_launch_us'], r.get('valu_active'))
print('sor', {k: (v['one_pair']['seconds'], v['batch']['ms_per_pair'], v['batch']['frac_of_hbm_peak']) for k, v in d['sor'].items()}); print('occ', d['occ']['one_triple'], d['occ']['batch'])"
