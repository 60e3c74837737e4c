cd $GRAFT_REPO_ROOT
A="--no-cpu --no-sor --no-occ --no-4k --no-cli --no-other-mode --fixed-steps 0"
for o in "fuse3=2" "fuse3=1" "fuse3=1 --opt fuse3_cursor=0"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 $A --opt $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'])"
done
