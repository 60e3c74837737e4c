#!/bin/bash
# round 2, GPU session 10: full suite on the current tree; A/B of the shared-denominator division variant
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02j
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout -k 10 900 tools/ab_div_shared.sh run > $OUT/ab_div_shared.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_div_shared.txt
