#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats of the same command + PMC traffic passes.
# Everything lands under gpurun_out/final/; tools/summarize_profiles.py turns it into the files under profiles/.
set -e
R=$GRAFT_REPO_ROOT
OUT=${OFX_PROF_OUT:-$R/gpurun_out/final}
TAG=${OFX_PROF_TAG:-r02_f}
mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py --steps 64 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
# BASELINE config 5 size (3840x2160) and the f32 fast mode, same command otherwise
timeout -k 10 400 python bench.py --steps 16 --warmup 1 --nx 3840 --ny 2160 --no-cpu --no-sor --no-occ --no-4k > $OUT/bench_4k.json 2> $OUT/bench_4k.err
timeout -k 10 400 python bench.py --steps 64 --warmup 2 --precision f32 --no-cpu --no-sor --no-occ --no-4k > $OUT/bench_f32.json 2> $OUT/bench_f32.err
timeout -k 10 400 python bench.py --steps 16 --warmup 1 --nx 3840 --ny 2160 --precision f32 --no-cpu --no-sor --no-occ --no-4k > $OUT/bench_4k_f32.json 2> $OUT/bench_4k_f32.err
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats of the SAME command (one pair in flight so that per-kernel times are not overlapped)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu --no-4k --no-sor --no-occ --streams 1 --lockstep 1 > $OUT/trace.json 2> $OUT/trace.err
# PMC passes (own runs, --kernel-trace only): HBM traffic of the dominant kernel at the bench size and at 4K
for sz in 1920x1080 3840x2160; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_fetch_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_write_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$sz -- python3 $R/tools/pmc_iter.py $sz n=40 > $OUT/pmc_sq_$sz.log 2>&1
done
# the same counters on the kernel as the job launches it: lockstep group of 16 (1080p) / 4 (4K) pairs per launch
for spec in 1920x1080:16 1920x1080:5 3840x2160:4; do
  sz=${spec%%:*}; g=${spec##*:}; [ $g = 5 ] && sz=${sz}_g5
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmcg_fetch_$sz -- python3 $R/tools/pmc_group.py ${sz%%_*} G=$g > $OUT/pmcg_fetch_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmcg_write_$sz -- python3 $R/tools/pmc_group.py ${sz%%_*} G=$g > $OUT/pmcg_write_$sz.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmcg_sq_$sz -- python3 $R/tools/pmc_group.py ${sz%%_*} G=$g > $OUT/pmcg_sq_$sz.log 2>&1
done
# kernel trace of the group launches alone (what bench.py's roofline object times with HIP events)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_group -- python3 $R/tools/pmc_group.py 1920x1080 G=16 > $OUT/trace_group.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_group5 -- python3 $R/tools/pmc_group.py 1920x1080 G=5 > $OUT/trace_group5.log 2>&1
# SOR window kernels in a lockstep group of 16 pairs (hyperplane-major layout): per-kernel stats + counters
for cfg in hs_cfg3 brox_cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sor_trace_$cfg -- python3 $R/tools/bench_sor_groups.py --only=$cfg --grid=1x16 > $OUT/sor_trace_$cfg.jsonl 2> $OUT/sor_trace_$cfg.err || true
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sor_pmc_mem_$cfg -- python3 $R/tools/bench_sor_groups.py --only=$cfg --grid=1x16 --no-warm > $OUT/sor_pmc_mem_$cfg.log 2>&1 || true
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sor_pmc_sq_$cfg -- python3 $R/tools/bench_sor_groups.py --only=$cfg --grid=1x16 --no-warm > $OUT/sor_pmc_sq_$cfg.log 2>&1 || true
  python3 - $OUT $cfg <<'PY' > $OUT/sor_counters_$cfg.txt 2>&1 || true
import collections, csv, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
pat = "k_hs_window" if cfg.startswith("hs") else "k_brox_window"
for kind in ("mem", "sq"):
    fs = glob.glob("%s/sor_pmc_%s_%s/**/*_counter_collection.csv" % (out, kind, cfg), recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    kt = glob.glob("%s/sor_pmc_%s_%s/**/*_kernel_trace.csv" % (out, kind, cfg), recursive=True)
    tot = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt[0])) if pat in r["Kernel_Name"]) / 1e9
    print(cfg, kind, "launches", max(n.values()) if n else 0, "kernel seconds %.4f" % tot, " ".join("%s_total=%.6g" % (k, v) for k, v in sorted(acc.items())))
PY
  cat $OUT/sor_counters_$cfg.txt
done
head -6 $OUT/sor_trace_hs_cfg3/*/*kernel_stats.csv | cut -c1-260 > $OUT/sor_kernel_stats.txt || true
head -8 $OUT/sor_trace_brox_cfg4/*/*kernel_stats.csv | cut -c1-260 >> $OUT/sor_kernel_stats.txt || true
cat $OUT/sor_trace_hs_cfg3.jsonl $OUT/sor_trace_brox_cfg4.jsonl
cd $R
OFX_PROF_SRC=$OUT OFX_PROF_DST=$OUT/summary python3 tools/summarize_profiles.py $TAG > $OUT/summary.log 2>&1; echo "summarize rc=$?"
tail -30 $OUT/summary.log
# the raw traces are large; the summaries above are what gets committed
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*_counter_collection.csv' -delete
cat $OUT/bench.json | cut -c1-600
