#!/bin/bash
# session 15: SQ counters of k_rof_window (what does a step spend its cycles on?)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02o; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" ; do
  i=$((i+1)); rm -rf $O/pmc$i
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -- python3 tools/bench_tvl1occ.py --size 640x480 --cpu none > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; }
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_rof_window" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print("%-24s launches %6d  mean per launch %14.1f" % (k, n, v / n))
PY
  find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
done
