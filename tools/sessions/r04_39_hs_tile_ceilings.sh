#!/bin/bash
# k_hs_tile (Horn-Schunck tolerance mode): production against its two ceilings at fixed work, batch 3 x 16 at 1920x1080:
#   mem = loads, LDS set-up and stores without the colour steps (-DOFX_HST_CEIL=1); alu = the colour steps without global loads / stores (-DOFX_HST_CEIL=2).
# The ceiling builds: git apply tools/patches/hs_tile_ceilings.patch && tools/build_variant.sh hst_mem -DOFX_HST_CEIL=1 && tools/build_variant.sh hst_alu -DOFX_HST_CEIL=2 && git apply -R tools/patches/hs_tile_ceilings.patch
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
{
for v in "" hst_mem hst_alu; do
  echo "== ${v:-production}, fixed work (150 sweeps per solve)"
  OFX_LIB_PATH=${v:+$PWD/variants/libofx_$v.so} timeout -k 10 400 python tools/bench_sor_groups.py --only=hs_cfg3 --grid=3x16 --opt=sor_exact=0 --opt=fixed_work=1 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r04_hs_tile_ceilings.txt 2>&1
cut -c1-330 gpurun_out/r04_hs_tile_ceilings.txt
