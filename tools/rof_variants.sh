#!/bin/bash
# Where does a step of k_rof_window go?  Variant builds (timing only, results of 1 and 3 are wrong):
#   1 noarith  : the inner cell's divisions and elimination replaced by four additions
#   2 nostore  : no global stores from the steps
#   3 nobarrier: no barrier between the steps
#   4 skeleton : window fill, step loop with its barriers, write-back -- no cells at all
#   5 readsonly: a cell = its LDS reads, nothing else
# Build in the container: tools/rof_variants.sh build     On the GPU box: tools/rof_variants.sh run
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  for v in 1 2 3 4 5; do
    make -s -C optical-flow-1_amd/csrc OUT=$R/variants/libofx_rof$v.so BUILD=$R/variants/build_rof$v EXTRA=-DROF_VAR=$v -j4
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $R
mkdir -p gpurun_out/rofvar
for v in 0 1 4 5; do
  [ $v = 0 ] && unset OFX_LIB_PATH || export OFX_LIB_PATH=$R/variants/libofx_rof$v.so
  rm -rf gpurun_out/rofvar/v$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rofvar/v$v -- python3 tools/bench_tvl1occ.py --size 640x480 --cpu none > gpurun_out/rofvar/v$v.jsonl 2> gpurun_out/rofvar/v$v.err || { echo "variant $v failed"; tail -3 gpurun_out/rofvar/v$v.err; exit 1; }
  find gpurun_out/rofvar/v$v -name "*kernel_trace.csv" -delete
  echo "== variant $v: $(find gpurun_out/rofvar/v$v -name '*kernel_stats.csv' | head -1 | xargs grep k_rof_window)"
done
