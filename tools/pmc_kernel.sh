#!/bin/bash
# tools/pmc_kernel.sh <kernel substring> <total grid threads or 0> "<counters pass 1>" "<counters pass 2>" ... :
# average PMC values per launch of one kernel over a short serial bench run (own rocprofv3 run per pass)
pat=$1; gx=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "$@"; do
  rm -rf /tmp/pmck
  timeout -k 10 250 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmck -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --fixed-steps 0 --streams 1 --lockstep 1 > /tmp/pmck.json 2> /tmp/pmck.err
  python3 - "$pat" "$gx" <<'PY'
import collections, csv, glob, sys
pat, gx = sys.argv[1], int(sys.argv[2])
f = glob.glob("/tmp/pmck/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"] and (gx == 0 or int(r["Grid_Size"]) == gx):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
kt = glob.glob("/tmp/pmck/**/*_kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))
     if pat in r["Kernel_Name"] and (gx == 0 or int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) == gx)]
print("launches", len(d), "avg_us %.2f" % (sum(d) / max(len(d), 1)), " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
