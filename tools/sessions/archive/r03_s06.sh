#!/bin/bash
# round 3 session 6: suite, the driver's command with the tolerance-mode headline (+ strict beside it), PMC passes of the group
# launches in both modes, trace + budget of the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions'], 'strict', d.get('strict'))
print('single', d['single_pair']); print('fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'roof4k', d['roofline_4k']['frac'], d['roofline_4k']['avg_launch_us'])
print('sor', {k: (v['one_pair']['seconds'], v['batch']['frac_of_hbm_peak']) for k, v in d['sor'].items()}); print('occ', d['occ']); print('cpu', d['cpu_baseline'])"
bash tools/pmc_round3.sh $O/pmc > $O/pmc.log 2>&1; tail -5 $O/pmc.log
cd /tmp && export TMPDIR=/tmp
export OFX_BENCH_MARK=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ --no-other-mode > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
python3 tools/trace_budget.py $O/trace > $O/budget.txt 2>&1; head -36 $O/budget.txt
(echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ --no-other-mode"; head -24 $O/trace/*/*kernel_stats.csv | cut -c1-260) > $O/kernel_stats_head.txt
find $O -name '*_kernel_trace.csv' -size +40M -delete
