#!/bin/bash
# round 2, GPU session 9: several sweeps per workgroup in the windowed SOR kernels -- parity, then throughput by sor_spw
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
OFX_FUZZ_SOR_GROUPS=12 OFX_FUZZ_SOR=8 timeout -k 10 900 python -m pytest tests/test_gpu_sor.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/sweep_sor_groups.py > $OUT/sor_sweep_spw.jsonl 2> $OUT/sor_sweep_spw.err; echo "sweep rc=$?"
cat $OUT/sor_sweep_spw.jsonl
