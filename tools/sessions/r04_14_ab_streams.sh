cd $GRAFT_REPO_ROOT
A="--no-cpu --no-sor --no-occ --no-4k --no-cli --no-single --no-other-mode --fixed-steps 0"
for spec in "4 5" "2 10" "5 4" "3 7" "4 0" "1 20" "2 5"; do
  set -- $spec
  for r in 1 2; do
    python bench.py --gpus 1 --steps 20 --warmup 5 $A --streams $1 --lockstep $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 lockstep $2', d['value'], d['config'].get('lockstep_group'), d['config'].get('streams_per_gpu'))"
  done
done
