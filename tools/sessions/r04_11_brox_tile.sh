cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sor_tile.py -x -q -m gpu -k "brox or f32" 2>&1 | tail -5
python tools/check_sor_tile.py 2>&1 | grep -E "MISMATCH|cfg4|MISMATCHES" | cut -c1-230
for K in 9 1 2 4; do echo "== K $K"; python tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16,3x16 --opt=sor_exact=0 --opt=sor_fuse=$K 2>&1 | grep config | cut -c1-250; done
