#!/bin/bash
# round 3 session 3: common-domain hypot (unscaled sqrt / division expansions), spin-wait polls; A/B against session 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
for tag in default spin0; do
  opt=""; [ $tag = spin0 ] && opt="--opt spin_us=0"
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ $opt > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"
  python3 -c "import json; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag value', d['value'], d['repetitions']['seconds'], 'single', d['single_pair']['device_resident']['ms_per_pair'], d['single_pair']['host_entry']['ms_per_pair'], 'fixed', d['fixed_work']['value'], d['fixed_work']['single_pair'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['single_pair']['avg_launch_us'], 'roof4k', d['roofline_4k']['frac'], d['roofline_4k']['avg_launch_us'])"
done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --opt relaxed_dual=1 --no-cpu --no-sor --no-occ > $O/bench_relaxed.json 2> $O/bench_relaxed.err; echo "bench relaxed rc=$?"
python3 -c "import json; d=json.loads(open('$O/bench_relaxed.json').read().strip().splitlines()[-1]); print('relaxed value', d['value'], 'single', d['single_pair']['device_resident']['ms_per_pair'], 'fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'roof4k', d['roofline_4k']['frac'], d['roofline_4k']['avg_launch_us'])"
