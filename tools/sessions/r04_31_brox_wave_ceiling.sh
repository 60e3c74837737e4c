#!/bin/bash
# k_brox_wave (Brox tolerance mode, finest level): production against its memory ceiling (update replaced by a copy), fixed work, batch 3 x 16.
# The ceiling build: git apply tools/patches/brox_wave_ceil_mem.patch && tools/build_variant.sh brw_mem -DOFX_CEIL_MEM && git apply -R tools/patches/brox_wave_ceil_mem.patch
mkdir -p gpurun_out
{
echo "== production, fixed work"
timeout -k 10 300 python tools/bench_sor_groups.py --only=brox_cfg4 --grid=3x16 --opt=sor_exact=0 --opt=fixed_work=1 2>&1 | grep -v amdgpu.ids
echo "== memory ceiling, fixed work"
OFX_LIB_PATH=$PWD/variants/libofx_brw_mem.so timeout -k 10 300 python tools/bench_sor_groups.py --only=brox_cfg4 --grid=3x16 --opt=sor_exact=0 --opt=fixed_work=1 2>&1 | grep -v amdgpu.ids
echo "== production, data-dependent stop"
timeout -k 10 300 python tools/bench_sor_groups.py --only=brox_cfg4 --grid=3x16 --opt=sor_exact=0 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r04_brox_wave_ceiling.txt 2>&1
cut -c1-330 gpurun_out/r04_brox_wave_ceiling.txt
