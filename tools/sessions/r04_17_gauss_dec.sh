cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tvl1.py tests/test_gpu_sor.py tests/test_gpu_occ.py -x -q -m gpu 2>&1 | tail -4
python tools/ab_bench.py dec=,gauss_fused=1 nodec=,gauss_fused=3 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-cli --no-single --no-other-mode --fixed-steps 0" 2>&1 | tail -3
