#!/bin/bash
# round 3 session 20: where one 640x480 / 1080p triple's time goes with the ROF iterations in flight (kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sz in 640x480 1920x1080; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/occ_$sz -- python3 $R/tools/bench_tvl1occ.py --size $sz --cpu none > $O/trace_$sz.log 2>&1 || { tail -5 $O/trace_$sz.log; exit 1; }
  grep '"size"' $O/trace_$sz.log
  (echo "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_tvl1occ.py --size $sz --cpu none (warm solve of 1 level + the timed solve)"; cut -c1-200 /tmp/occ_$sz/*/*kernel_stats.csv | head -30) > $O/kernel_stats_$sz.txt
  cat $O/kernel_stats_$sz.txt | cut -c1-160
  python3 $R/tools/trace_budget.py /tmp/occ_$sz --all > $O/budget_$sz.txt 2>&1; tail -25 $O/budget_$sz.txt | cut -c1-200
done
