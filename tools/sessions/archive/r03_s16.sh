#!/bin/bash
# round 3 session 16: the two Horn-Schunck window kernels (global round trips / LDS window) in a lockstep group of 16 under the
# kernel trace and three counter passes -- what bounds a step when the chip is shared
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r03p; mkdir -p $W
CMD="$R/tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x16 --no-warm"
pass() {   # tag, rocprofv3 arguments
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $W/$tag -- python3 $CMD --opt=sor_lds=$V > $W/$tag.log 2>&1
  local rc=$?
  echo "== sor_lds=$V pass $tag rc=$rc"
  if [ $rc -ne 0 ]; then tail -4 $W/$tag.log | cut -c1-300; return; fi
  python3 $R/tools/pmc_sum.py $W/$tag k_hs_window
  grep '"config"' $W/$tag.log | cut -c1-220
  rm -rf $W/$tag
}
for V in 0 2; do
  pass trace --kernel-trace --stats
  pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace
  pass mem --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace
  pass lds --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace
done 2>&1 | tee $O/summary.txt
