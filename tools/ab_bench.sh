#!/bin/bash
# A/B of library builds with the real bench on one device: tools/ab_bench.sh name=path ...  (two interleaved rounds)
for round in 1 2; do
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  OFX_LIB_PATH=$path timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'value', d['value'], 'fixed4', d['fixed_work']['value'], 'single', d['fixed_work']['single_pair']['value'], [l['iter_us'] for l in d['fixed_work']['levels']], 'frac', d['roofline']['frac'])"
done
done
