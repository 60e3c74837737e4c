#!/bin/bash
# round 3 session 17: idle (sweep, block) units leave at once; HS LDS window with register operands (sor_lds=3): parity, then the
# group regime for the three kernels and two row-block heights, and the lone solve
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sor.py -m gpu -x -q > $O/sor_tests.log 2>&1; rc=$?; echo "sor tests rc=$rc"; tail -4 $O/sor_tests.log
[ $rc -ne 0 ] && exit 1
for v in 0 2 3; do for rows in 125 61; do
  echo "== sor_lds=$v rows=$rows"
  timeout -k 10 200 python tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x1,1x16,2x16 --opt=sor_lds=$v --opt=sor_rows=$rows 2>&1 | cut -c1-230 || exit 1
done; done | tee $O/hs_groups.txt
