#!/bin/bash
# Strict mode (the library default): three iterations per launch forced (fuse3 = 1) against the default two, on the aligned kernel.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
A="--no-cpu --no-sor --no-occ --no-4k --no-cli --no-other-mode --no-single --mode strict"
for r in 1 2; do for o in "fuse3=2" "fuse3=1"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 $A --opt $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['fixed_work']['value'], d['roofline']['avg_launch_us'], d['roofline'].get('iterations_per_launch'))"
done; done > gpurun_out/r04_strict_fuse3.txt 2>&1
cat gpurun_out/r04_strict_fuse3.txt
