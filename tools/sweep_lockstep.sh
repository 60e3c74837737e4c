#!/bin/bash
# tools/sweep_lockstep.sh st,ls,cc[,extra args...] ...: headline value vs (streams, lockstep, concurrency hint) on one device
for cfg in "$@"; do
  IFS=, read st ls cc extra <<< "$cfg"
  timeout -k 10 300 python bench.py --steps 64 --warmup 2 --no-cpu --fixed-steps 1 --streams $st --lockstep $ls --concurrency $cc $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $st lockstep $ls conc $cc $extra', 'value', d['value'], 'ms/step', d['ms_per_step'], 'fixed', d['fixed_work']['value'], 'frac', d['roofline']['frac'])"
done
