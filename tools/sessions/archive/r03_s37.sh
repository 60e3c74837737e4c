#!/bin/bash
# round 3 session 37: final-tree record -- full GPU suite, smoke, soak with three iterations per launch forced at every size, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ak; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -4 $O/suite.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
OFX_FUZZ_OPTS="fuse3=1" OFX_FUZZ_SEED=211 OFX_FUZZ_N=120 OFX_FUZZ_SOR=4 OFX_FUZZ_GROUPS=24 OFX_FUZZ_TEMPORAL=1 OFX_FUZZ_SOR_GROUPS=2 OFX_FUZZ_OCC=10 OFX_FUZZ_REXPO=2 \
  timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz_fuse3.log 2>&1; echo "soak rc=$?"; tail -3 $O/fuzz_fuse3.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions']['seconds'], 'strict', d['strict']['value'])
print('single', d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'], 'fixed', d['fixed_work']['value'])
r=d['roofline']; print('roof', r['kernel'], r['frac'], r['avg_launch_us'], r.get('traffic'), r.get('valu_active'), r.get('two_iterations_per_launch'))
r=d['roofline_4k']; print('roof4k', r['kernel'], r['frac'], r['avg_launch_us'], r.get('two_iterations_per_launch'))
print('sor', {k: (v['one_pair']['seconds'], v['batch']['ms_per_pair'], v['batch']['frac_of_hbm_peak']) for k, v in d['sor'].items()}); print('occ', d['occ']['one_triple'], d['occ']['batch'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_vs_this_reference_run'])"
