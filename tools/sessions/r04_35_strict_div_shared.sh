#!/bin/bash
# Strict mode: the four IEEE quotients of the dual update with the reciprocal refinement shared per denominator (-DOFX_DIV_SHARED, bit-identical) against the compiler's divisions.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
A="--no-cpu --no-sor --no-occ --no-4k --no-cli --no-other-mode --no-single --mode strict"
for r in 1 2 3; do for lib in "" variants/libofx_divshared.so; do
  OFX_LIB_PATH=${lib:+$PWD/$lib} python bench.py --gpus 1 --steps 20 --warmup 5 $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-default}', d['value'], d['fixed_work']['value'], d['roofline']['avg_launch_us'])"
done; done > gpurun_out/r04_strict_div_shared.txt 2>&1
cat gpurun_out/r04_strict_div_shared.txt
python -m pytest tests/test_gpu_tvl1.py -x -q -m gpu -k "bitexact or strict or golden" > gpurun_out/r04_pytest_divshared.txt 2>&1; tail -3 gpurun_out/r04_pytest_divshared.txt
