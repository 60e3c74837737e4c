#!/bin/bash
# round 3 session 2: batched level set-up (pyramid / grad_pack / zoom_in / to_flo over the group), relaxed_dual accuracy + rate
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python3 -c "import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); print('value', d['value'], d['repetitions'], d['single_pair'], 'fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --opt relaxed_dual=1 --no-cpu --no-sor --no-occ > $O/bench_relaxed.json 2> $O/bench_relaxed.err; echo "bench relaxed rc=$?"
python3 -c "import json; d=json.loads(open('$O/bench_relaxed.json').read().strip().splitlines()[-1]); print('relaxed value', d['value'], d['repetitions'], 'fixed', d['fixed_work']['value'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'roof4k', d['roofline_4k']['frac'], d['roofline_4k']['avg_launch_us'])"
timeout -k 10 300 python tools/relaxed_accuracy.py > $O/relaxed_accuracy.jsonl 2> $O/relaxed_accuracy.err; echo "acc rc=$?"; cat $O/relaxed_accuracy.jsonl
cd /tmp && export TMPDIR=/tmp
export OFX_BENCH_MARK=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-sor --no-occ > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
python3 tools/trace_budget.py $O/trace > $O/budget.txt 2>&1; head -40 $O/budget.txt
find $O -name '*_kernel_trace.csv' -size +40M -delete
