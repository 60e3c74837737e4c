cd $GRAFT_REPO_ROOT
python tools/check_sor_tile.py > gpurun_out/r04_tile10.txt 2>&1; grep -c ": ok" gpurun_out/r04_tile10.txt; grep -E "MISMATCH|cfg4|Error|error" gpurun_out/r04_tile10.txt | cut -c1-200 | head
python tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16,3x16 --opt=sor_exact=0 2>&1 | cut -c1-250
bash tools/sessions/r04_05_cli_budget.sh 2>&1 | grep -E "real|phases_ms" | cut -c1-200
