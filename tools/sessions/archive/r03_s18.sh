#!/bin/bash
# round 3 session 18: memory-side counters of the three HS window kernels in a lockstep group of 16 (one scale, two warps: ~1500
# launches, the TCC counters cost ~10 ms per dispatch to read)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r03r; mkdir -p $W
CMD="$R/tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x16 --no-warm --kw=nscales=1 --kw=warps=2"
pass() {
  local tag=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $W/$tag -- python3 $CMD --opt=sor_lds=$V > $W/$tag.log 2>&1
  local rc=$?
  echo "== sor_lds=$V pass $tag rc=$rc"
  if [ $rc -ne 0 ]; then tail -4 $W/$tag.log | cut -c1-300; return; fi
  python3 $R/tools/pmc_sum.py $W/$tag k_hs_window
  grep '"config"' $W/$tag.log | cut -c1-260
  rm -rf $W/$tag
}
for V in 0 2 3; do
  pass trace --kernel-trace --stats
  pass mem --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace
  pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --kernel-trace
  pass tcp --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --kernel-trace
done 2>&1 | tee $O/summary.txt
