#!/bin/bash
# round 3 session 28: full GPU suite, smoke, the driver's command, counter passes of the three-iteration launches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ab; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -6 $O/suite.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions'], 'strict', d.get('strict'))
print('single', d['single_pair']); print('fixed', d['fixed_work']['value'])
r=d['roofline']; print('roof', r['kernel'], r['frac'], r['avg_launch_us'], r.get('two_iterations_per_launch'), r.get('mpix_iters_per_s'))
r=d['roofline_4k']; print('roof4k', r['kernel'], r['frac'], r['avg_launch_us'], r.get('two_iterations_per_launch'), r.get('mpix_iters_per_s'))
print('sor', {k: (v['one_pair']['seconds'], v['batch']['frac_of_hbm_peak']) for k, v in d['sor'].items()}); print('occ', d['occ']); print('cpu', d['cpu_baseline'])"
ONLY_ITER3=1 bash tools/pmc_round3.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log; python3 -c "
import json; d=json.load(open('$O/pmc/summary.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('counter_tb_per_s','traffic_over_fused_compulsory','valu_active_fraction','rocprofv3_stats_avg_us','frac_of_8tbs','l2_hit_rate','wait_any_over_wave_cycles','wait_inst_over_wave_cycles')})"
