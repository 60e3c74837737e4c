#!/bin/bash
# round 3 session 10: suite after the merged loop memset / occ fixes / docs-time code changes, smoke, driver command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions']['seconds'], 'strict', d['strict']['value'])
print('single', d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'])
print('fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('hbm_frac_counter'), 'roof4k', d['roofline_4k']['frac'], d['roofline_4k']['avg_launch_us'])
print('sor', {k: (v['one_pair']['seconds'], v['batch']['frac_of_hbm_peak']) for k, v in d['sor'].items()}); print('occ', d['occ']['one_triple'], d['occ']['batch']['ms_per_triple'])"
timeout -k 10 400 python bench.py --gpus 1 --workload 4k-batch --steps 64 --warmup 1 --no-cpu --no-sor --no-occ > $O/bench_4kbatch.json 2> $O/bench_4kbatch.err; echo "4k-batch rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_4kbatch.json').read().strip().splitlines()[-1])
print('4k-batch value', d['value'], d['repetitions']['seconds'], 'strict', d['strict']['value'], 'roof', d['roofline']['frac'])"
