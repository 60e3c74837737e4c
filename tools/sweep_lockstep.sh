#!/bin/bash
# tools/sweep_lockstep.sh: headline value vs (streams, lockstep, concurrency hint) on one device
for cfg in "$@"; do
  IFS=, read st ls cc <<< "$cfg"
  timeout -k 10 300 python bench.py --steps 32 --warmup 2 --no-cpu --streams $st --lockstep $ls --concurrency $cc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $st lockstep $ls conc $cc', 'value', d['value'], 'ms/step', d['ms_per_step'], 'fixed', d['fixed_work']['value'], 'single', d['fixed_work']['single_pair']['value'], [l['iter_us'] for l in d['fixed_work']['levels']], 'frac', d['roofline']['frac'])"
done
