#!/bin/bash
# Horn-Schunck tolerance mode, one pair at a time: tile geometry x sweeps per launch (results do not depend on either).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for geo in 1 2 3; do for k in 2 3 4; do
  python tools/sor_one_pair.py hs --exact=0 --fuse=$k --reps=5 sor_tile=$geo 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('geometry $geo K $k', d['seconds_min'], d['seconds'])"
done; done > gpurun_out/r04_hs_lone_pair_geometry.txt 2>&1
cat gpurun_out/r04_hs_lone_pair_geometry.txt
