#!/bin/bash
# round 3 session 23: alfa stage inside the sweep's workgroup for lockstep groups: parity (groups), batches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03w; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_occ.py -m gpu -x -q > $O/occ_tests.log 2>&1; rc=$?; echo "occ tests rc=$rc"; tail -5 $O/occ_tests.log
[ $rc -ne 0 ] && exit 1
OFX_FUZZ_SEED=78 OFX_FUZZ_OCC=30 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "occ" > $O/occ_fuzz.log 2>&1; rc=$?; echo "occ fuzz rc=$rc"; tail -5 $O/occ_fuzz.log
[ $rc -ne 0 ] && exit 1
for spec in "1:16 16" "2:32 16" "4:32 8" "4:64 16" "2:14 7 1920x1080"; do
  set -- $spec
  sz=${3:-640x480}
  timeout -k 10 300 python tools/bench_tvl1occ.py --size $sz --cpu none --batch $1 --opt lockstep=$2 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'size': d['size'], 'options': d['options'], 'one_triple_s': d['gpu_s'], 'batch': d['batch']}))" || exit 1
done | tee $O/occ_batches.txt
