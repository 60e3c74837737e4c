# time budget of the driver's command (tools/trace_budget.py): kernel trace of bench.py --gpus 1 --steps 20 --warmup 5 with region markers
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_budget
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OFX_BENCH_MARK=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ --no-cli --no-4k --no-single --no-other-mode --fixed-steps 0 > $OUT/bench.json 2> $OUT/bench.err
cd $R
python3 tools/trace_budget.py $OUT/trace > $OUT/time_budget.txt 2>&1
f=$(find $OUT/trace -name '*_kernel_stats.csv' | head -1); [ -n "$f" ] && python3 tools/fmt_kernel_stats.py $OUT/trace > $OUT/kernel_stats.txt 2>/dev/null
find $OUT -name '*_kernel_trace.csv' -delete
head -45 $OUT/time_budget.txt
