#!/bin/bash
# like ab_bench.sh but prints the full-resolution iteration time at 1080p and 4K (single pair, fixed work)
for round in 1 2; do
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  for sz in "1920 1080" "3840 2160"; do set -- $sz
  OFX_LIB_PATH=$path timeout -k 10 200 python bench.py --nx $1 --ny $2 --steps 4 --warmup 1 --no-cpu --streams 1 --fixed-steps 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$1x$2', [l['iter_us'] for l in d['fixed_work']['levels']], 'frac', d['roofline']['frac'])"
  done
done
done
