#!/bin/bash
# round 3 session 13: record run -- suite, the driver's exact command plain and under rocprofv3 --kernel-trace --stats (+ budget)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], d['config']['arithmetic_mode'], d['repetitions']['seconds'], 'strict', d['strict']['value'], 'single', d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'])"
cd /tmp && export TMPDIR=/tmp
export OFX_BENCH_MARK=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
python3 tools/trace_budget.py $O/trace > $O/budget.txt 2>&1; head -34 $O/budget.txt
(echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5      (MI355X; the whole run: warm-up, 5 timed repetitions per f64 mode,"; echo "# single-pair / fixed-work / roofline / sor / occ / cpu legs; value under the profiler: $(python3 -c "import json; print(json.loads(open('$O/trace_bench.json').read().strip().splitlines()[-1])['value'])"))"; head -40 $O/trace/*/*kernel_stats.csv | cut -c1-300) > $O/driver_command_kernel_stats.txt
find $O -name '*_kernel_trace.csv' -size +40M -delete
