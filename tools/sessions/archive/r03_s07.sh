#!/bin/bash
# round 3 session 7: fused Gaussian (suite + A/B on one box), PCIe pinning experiment
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log
timeout -k 10 120 python tools/pcie_pinning.py > $O/pcie_pinning.txt 2>&1; cat $O/pcie_pinning.txt
timeout -k 10 600 python tools/ab_bench.py fused= twopass=,gauss_fused=0 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-other-mode --no-single" > $O/ab.txt 2>&1; cat $O/ab.txt
