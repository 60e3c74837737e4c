#!/bin/bash
# Round-3 PMC passes on the iteration kernel as the job launches it (lockstep groups), strict and tolerance mode: HBM-side
# traffic (FETCH_SIZE / WRITE_SIZE in separate passes, MI355X_MICROARCH.md "HBM"), SQ activity, and the kernel stats of the same
# launches.  Run on the GPU box; tools/pmc_summary_r03.py turns the output into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/r03pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in 1920x1080:5 3840x2160:4; do
  sz=${spec%%:*}; g=${spec##*:}
  for v in 0:0 1:0 1:1; do                     # relaxed_dual : fuse3  (strict two-iteration, tolerance two-iteration, tolerance three-iteration kernel)
    m=${v%%:*}; f=${v##*:}
    tag=${sz}_g${g}_m${m}; [ $f = 1 ] && tag=${tag}i3
    [ -n "$ONLY_ITER3" ] && [ $f != 1 ] && continue
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/pmc_group.py $sz G=$g relaxed=$m fuse3=$f > $OUT/fetch_$tag.log 2>&1
    timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/write_$tag -- python3 $R/tools/pmc_group.py $sz G=$g relaxed=$m fuse3=$f > $OUT/write_$tag.log 2>&1
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq_$tag -- python3 $R/tools/pmc_group.py $sz G=$g relaxed=$m fuse3=$f > $OUT/sq_$tag.log 2>&1
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- python3 $R/tools/pmc_group.py $sz G=$g relaxed=$m fuse3=$f > $OUT/stats_$tag.log 2>&1
  done
done
cd $R
python3 tools/pmc_summary_r03.py $OUT > $OUT/summary.json 2> $OUT/summary.err; cat $OUT/summary.json | head -80
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_counter_collection.csv' -delete
