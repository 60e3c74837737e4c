#!/bin/bash
# round 3 session 30: radius-templated fused Gaussian: parity, A/B on the driver's command; the 4k-batch workload on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ad; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_tvl1.py tests/test_gpu_ops.py -m gpu -x -q -k "strips or fuse3 or not tile" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python tools/ab_bench.py "gauss_generic=,gauss_fused=2" "gauss_templated=" --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single --fixed-steps 0" 2>&1 | tee $O/ab.txt
timeout -k 10 600 python bench.py --workload 4k-batch --gpus 1 --no-cpu --no-sor --no-occ > $O/bench_4k_batch.json 2> $O/bench_4k_batch.err; echo "4k-batch rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_4k_batch.json').read().strip().splitlines()[-1])
print('4k-batch value', d['value'], d['config'].get('arithmetic_mode'), d.get('repetitions'), 'strict', (d.get('strict') or {}).get('value'), 'ms_per_step', d['ms_per_step'])"
