#!/bin/bash
# round 3 session 29: the driver's exact command under rocprofv3 --kernel-trace --stats with the three-iteration kernel as the default + time budget
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ac; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export OFX_BENCH_MARK=1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_ac -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
python3 tools/trace_budget.py /tmp/trace_ac > $O/budget.txt 2>&1; head -40 $O/budget.txt
(echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5      (MI355X; the whole run: warm-up, 5 timed repetitions per f64 mode,"; echo "# single-pair / fixed-work / roofline / sor / occ / cpu legs; value under the profiler: $(python3 -c "import json; print(json.loads(open('$O/trace_bench.json').read().strip().splitlines()[-1])['value'])"))"; python3 tools/fmt_kernel_stats.py /tmp/trace_ac 40) > $O/driver_command_kernel_stats.txt
