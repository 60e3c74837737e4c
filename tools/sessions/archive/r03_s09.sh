#!/bin/bash
# round 3 session 9: where a single pair's 5.6 ms go (kernel trace of device-resident single-pair solves)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/single_pair_trace.py tolerance > $O/run.txt 2>&1; tail -2 $O/run.txt
cd $R
python3 tools/trace_budget.py $O/trace > $O/budget.txt 2>&1; cat $O/budget.txt
for v in "" "spin_us=0" "chunk=4" "chunk=16"; do echo "== $v"; timeout -k 10 100 python3 tools/single_pair_trace.py tolerance $v 2>&1 | grep "single pair" | cut -c1-40; done
find $O -name '*_kernel_trace.csv' -size +40M -delete
