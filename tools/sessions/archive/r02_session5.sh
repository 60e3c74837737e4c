#!/bin/bash
# round 2, GPU session 5: full suite (new occ / colour tests), SOR geometry sweep on the skewed layout, memory-budget check,
# bench with loop-end statistics, store_a A/B at 4K
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/sweep_sor_groups.py > $OUT/sor_sweep.jsonl 2> $OUT/sor_sweep.err; echo "sor sweep rc=$?"
cat $OUT/sor_sweep.jsonl
timeout -k 10 600 python tools/bench_sor_groups.py --check > $OUT/sor_groups.jsonl 2> $OUT/sor_groups.err; echo "sor groups rc=$?"
cat $OUT/sor_groups.jsonl
timeout -k 10 400 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print('value',d['value'],'loop_ends',d.get('loop_ends'),'sor',{k:(v['mpix_sweeps_per_s'],v['seconds']) for k,v in d['sor'].items()})"
for sa in 1 0; do
  timeout -k 10 400 python bench.py --no-cpu --no-sor --no-4k --nx 3840 --ny 2160 --steps 16 --warmup 1 --opt store_a=$sa > $OUT/bench4k_sa$sa.json 2> $OUT/bench4k_sa$sa.err; echo "bench4k store_a=$sa rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench4k_sa$sa.json'));print('4K store_a=$sa value',d['value'],'loop_ends',d.get('loop_ends'))"
done
