#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats of the same command + PMC traffic passes.
# Everything lands under gpurun_out/final/; tools/summarize_profiles.py turns it into the files under profiles/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py --steps 64 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
# BASELINE config 5 size (3840x2160) and the f32 fast mode, same command otherwise
timeout -k 10 400 python bench.py --steps 16 --warmup 1 --nx 3840 --ny 2160 --no-cpu > $OUT/bench_4k.json 2> $OUT/bench_4k.err
timeout -k 10 400 python bench.py --steps 64 --warmup 2 --precision f32 --no-cpu > $OUT/bench_f32.json 2> $OUT/bench_f32.err
timeout -k 10 400 python bench.py --steps 16 --warmup 1 --nx 3840 --ny 2160 --precision f32 --no-cpu > $OUT/bench_4k_f32.json 2> $OUT/bench_4k_f32.err
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats of the SAME command (one pair in flight so that per-kernel times are not overlapped)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu --streams 1 --lockstep 1 > $OUT/trace.json 2> $OUT/trace.err
# PMC passes (own runs, --kernel-trace only): HBM traffic of the dominant kernel at the bench size and at 4K
for sz in 1920x1080 3840x2160; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_fetch_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_write_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_sq_$sz.log 2>&1
done
cat $OUT/bench.json
