#!/bin/bash
# round 2, GPU session 8: where the GPU time of the headline job goes (64 pairs, 4 contexts x groups of 16, eps = 0.01 only)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu --no-sor --no-4k --fixed-steps 0 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
python3 $R/tools/trace_busy.py $OUT/trace > $OUT/busy.txt 2>&1 || true
head -40 $OUT/busy.txt
find $OUT -name "*kernel_trace.csv" -delete
head -30 $OUT/trace/*/*kernel_stats.csv | cut -c1-200
