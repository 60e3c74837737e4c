#!/bin/bash
# round 3 session 1: baseline -- GPU suite, the driver's bench command, kernel trace of the same command + time budget
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
cut -c1-400 $O/bench_driver.json
cd /tmp && export TMPDIR=/tmp
export OFX_BENCH_MARK=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
python3 tools/trace_budget.py $O/trace > $O/budget.txt 2>&1; cat $O/budget.txt
head -25 $O/trace/*/*kernel_stats.csv | cut -c1-200 > $O/kernel_stats_head.txt
find $O -name '*_kernel_trace.csv' -size +40M -delete
