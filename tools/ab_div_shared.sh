#!/bin/bash
# A/B of the shared-denominator division in the TV-L1 dual update: variant build -DOFX_DIV_SHARED.
#   tools/ab_div_shared.sh build     (container)        tools/ab_div_shared.sh run   (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  make -s -C optical-flow-1_amd/csrc OUT=$R/variants/libofx_div_shared.so BUILD=$R/variants/build_div_shared EXTRA=-DOFX_DIV_SHARED -j4
  exit 0
fi
echo "== parity of the variant (kernel-level bit-exactness, multiscale, lockstep groups)"
OFX_LIB_PATH=$R/variants/libofx_div_shared.so timeout -k 10 600 python -m pytest tests/test_gpu_tvl1.py -m gpu -x -q 2>&1 | tail -3
echo "== production"
timeout -k 10 300 python tools/tune_iter.py --quick 2>&1 | grep -v amdgpu.ids
echo "== shared-denominator division"
OFX_LIB_PATH=$R/variants/libofx_div_shared.so timeout -k 10 300 python tools/tune_iter.py --quick 2>&1 | grep -v amdgpu.ids
for v in production div_shared; do
  L=$R/optical-flow-1_amd/libofx.so; [ $v = div_shared ] && L=$R/variants/libofx_div_shared.so
  OFX_LIB_PATH=$L timeout -k 10 300 python bench.py --no-cpu --no-sor --no-4k > /tmp/ab_$v.json 2> /tmp/ab_$v.err
  python -c "import json;d=json.load(open('/tmp/ab_$v.json'));print('$v: value',d['value'],'fixed',d['fixed_work']['value'],'launch us',d['roofline']['avg_launch_us'])"
done
