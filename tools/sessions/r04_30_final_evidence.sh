#!/bin/bash
# Final evidence of round 4 on one box: counter passes of the TV-L1 group launches (source hash recorded), the driver's command
# under the kernel trace (time budget), the full -m gpu suite, the driver's bench command, smoke().
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
bash tools/pmc_round3.sh $R/gpurun_out/r04pmc3 > gpurun_out/r04pmc3.log 2>&1
echo "pmc done: $(grep -c launch_us_counter_pass gpurun_out/r04pmc3/summary.json) entries"
GRAFT_REPO_ROOT=$R bash tools/sessions/r04_09_driver_budget.sh > gpurun_out/r04_budget.log 2>&1
echo "budget done"; grep -A5 "class shares" gpurun_out/r04_budget/time_budget.txt
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r04_pytest_final.txt 2>&1; tail -7 gpurun_out/r04_pytest_final.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; tail -2 gpurun_out/r04_bench_final.err
python -c "import __graft_entry__ as g; g.smoke()"
