#!/bin/bash
# ALU and memory ceilings of k_tvl1_iter2: two variant builds of libofx.so, each run once on the
# 4K and 1080p fixed-iteration passes of tools/tune_iter.py (HIP events around 100 iterations).
#   alu : arithmetic kept, loads alternate between two cache-resident rows of the strip, stores dropped   (-DOFX_CEIL_ALU)
#   mem : loads / stores / lane shifts kept, primal + dual arithmetic replaced by a copy                  (-DOFX_CEIL_MEM)
# Build the variants in the container first:  tools/ceilings.sh build      Then on the GPU box:  tools/ceilings.sh run
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  for v in alu mem; do
    V=$(echo $v | tr a-z A-Z)
    make -s -C optical-flow-1_amd/csrc OUT=$R/variants/libofx_ceil_$v.so BUILD=$R/variants/build_ceil_$v EXTRA=-DOFX_CEIL_$V -j4
  done
  exit 0
fi
echo "== production"
timeout -k 10 300 python tools/tune_iter.py --quick 2>&1 | grep -v amdgpu.ids
for v in alu mem; do
  echo "== ceiling $v"
  OFX_LIB_PATH=$R/variants/libofx_ceil_$v.so timeout -k 10 300 python tools/tune_iter.py --quick 2>&1 | grep -v amdgpu.ids
done
