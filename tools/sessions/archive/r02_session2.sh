#!/bin/bash
# round 2, GPU session 2: SOR lockstep groups -- parity tests, throughput grid, rocprofv3 per-kernel stats
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sor.py tests/test_gpu_api_errors.py tests/test_gpu_golden_cli.py -m gpu -x -q > $OUT/pytest_sor.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest_sor.log
tail -15 $OUT/pytest_sor.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/bench_sor_groups.py --check > $OUT/sor_groups.jsonl 2> $OUT/sor_groups.err; echo "sor groups rc=$?"
cat $OUT/sor_groups.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hs -- python3 $R/tools/bench_sor_groups.py --only=hs_cfg3 > $OUT/trace_hs.jsonl 2> $OUT/trace_hs.err; echo "trace hs rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_brox -- python3 $R/tools/bench_sor_groups.py --only=brox_cfg4 > $OUT/trace_brox.jsonl 2> $OUT/trace_brox.err; echo "trace brox rc=$?"
