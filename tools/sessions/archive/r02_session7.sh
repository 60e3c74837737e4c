#!/bin/bash
# round 2, GPU session 7: new API tests + CLI tests; how to spread the driver's 20 pairs over contexts / groups
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_api_errors.py tests/test_gpu_golden_cli.py tests/test_gpu_bench.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
for cfg in "4 0" "2 0" "3 0" "1 0" "5 0" "2 16" "4 8"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-sor --no-4k --fixed-steps 0 --streams $1 --lockstep $2 > $OUT/b20_$1_$2_$rep.json 2> $OUT/b20_$1_$2_$rep.err
    python -c "import json;d=json.load(open('$OUT/b20_$1_$2_$rep.json'));print('steps 20 streams $1 lockstep $2 ->', d['config']['lockstep_group'], 'value', d['value'])"
  done
done
for cfg in "4 0" "2 0" "3 0" "8 0"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 64 --warmup 2 --no-cpu --no-sor --no-4k --fixed-steps 0 --streams $1 --lockstep $2 > $OUT/b64_$1_$2.json 2> $OUT/b64_$1_$2.err
  python -c "import json;d=json.load(open('$OUT/b64_$1_$2.json'));print('steps 64 streams $1 lockstep $2 ->', d['config']['lockstep_group'], 'value', d['value'])"
done
