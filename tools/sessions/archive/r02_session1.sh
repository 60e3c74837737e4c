#!/bin/bash
# round 2, first GPU session: full -m gpu suite (with the new full-size tests), the driver's bench commands,
# kernel ceilings, config 5 on one GPU, rocprofv3 per-kernel stats of the SOR configs.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench64 rc=$?"
timeout -k 10 600 tools/ceilings.sh run > $OUT/ceilings.txt 2>&1; echo "ceilings rc=$?"
timeout -k 10 600 python bench.py --workload 4k-batch --no-cpu --no-sor --fixed-steps 1 --warmup 1 > $OUT/bench_4k_batch.json 2> $OUT/bench_4k_batch.err; echo "4k-batch rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sor_trace -- python3 $R/tools/bench_sor.py --no-per-step > $OUT/sor_trace.jsonl 2> $OUT/sor_trace.err; echo "sor trace rc=$?"
cd $R
cat $OUT/bench_driver.json
