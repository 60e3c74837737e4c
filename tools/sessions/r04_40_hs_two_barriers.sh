#!/bin/bash
# k_hs_tile with two barriers per sweep (the colours of a pixel row back to back inside a wave) against four (variants/libofx_hs4b.so, -DOFX_HST_FOUR_BARRIERS): parity, then the batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sor_tile.py tests/test_gpu_sor.py -x -q -m gpu > gpurun_out/r04_pytest_hs2b.txt 2>&1; tail -4 gpurun_out/r04_pytest_hs2b.txt
{
for v in "" hs4b; do for fw in 0 1; do
  echo "== ${v:-two barriers} fixed_work=$fw"
  OFX_LIB_PATH=${v:+$PWD/variants/libofx_$v.so} timeout -k 10 400 python tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x1,3x16 --opt=sor_exact=0 --opt=fixed_work=$fw 2>&1 | grep -v amdgpu.ids
done; done
} > gpurun_out/r04_hs_two_barriers.txt 2>&1
cut -c1-330 gpurun_out/r04_hs_two_barriers.txt
