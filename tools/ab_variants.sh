#!/bin/bash
# A/B of library builds on one device: each variant runs tools/tune_iter.py (usage: tools/ab_variants.sh name=path ...)
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  echo "== $name"
  OFX_LIB_PATH=$path timeout -k 10 300 python tools/tune_iter.py --quick 2>&1 | grep -v amdgpu.ids
done
