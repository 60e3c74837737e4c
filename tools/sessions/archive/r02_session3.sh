#!/bin/bash
# round 2, GPU session 3: TV-L1 stored-intermediate-state (no redo) parity + bench; SOR group geometry sweep + kernel stats
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_tvl1.py tests/test_gpu_fuzz.py tests/test_gpu_golden_cli.py tests/test_gpu_shim.py -m gpu -x -q > $OUT/pytest_tvl1.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest_tvl1.log
tail -15 $OUT/pytest_tvl1.log
[ $rc -ne 0 ] && exit $rc
for sa in 1 0; do
  timeout -k 10 400 python bench.py --no-cpu --no-sor --no-4k --opt store_a=$sa > $OUT/bench_store_a$sa.json 2> $OUT/bench_store_a$sa.err; echo "bench store_a=$sa rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_store_a$sa.json'));print('store_a=$sa value',d['value'],'fixed',d['fixed_work']['value'],'launch us',d['roofline']['avg_launch_us'])"
done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-sor > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench20 rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_driver.json'));print('driver cmd value',d['value'])"
timeout -k 10 900 python tools/sweep_sor_groups.py > $OUT/sor_sweep.jsonl 2> $OUT/sor_sweep.err; echo "sor sweep rc=$?"
cat $OUT/sor_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hs -- python3 $R/tools/bench_sor_groups.py --only=hs_cfg3 > $OUT/trace_hs.jsonl 2> $OUT/trace_hs.err; echo "trace hs rc=$?"
find $OUT -name "*kernel_trace.csv" -delete
head -8 $OUT/trace_hs/*/*kernel_stats.csv | cut -c1-230
