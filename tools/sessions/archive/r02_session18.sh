#!/bin/bash
# session 18: where does the eps job's GPU time go, per pyramid level?  (kernel trace of the bench job alone, fixed-work legs off)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02r; mkdir -p $O; rm -rf $O/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 64 --warmup 2 --no-cpu --no-4k --no-sor --fixed-steps 0 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -1 $O/bench.json | cut -c1-300
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/per_level.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    key = (name[:28], g if "iter" in name else 0)
    a = acc[key]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in acc.values())
print("span %.1f ms, kernel time sum %.1f ms (4 streams overlap)" % ((t1 - t0) / 1e6, tot / 1e6))
for k, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%-30s grid %9d  launches %6d  avg %8.2f us  total %8.2f ms  %5.1f %%" % (k[0], k[1], n, d / n / 1e3, d / 1e6, 100.0 * d / tot))
PY
find $O -name "*kernel_trace.csv" -delete
