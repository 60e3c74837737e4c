#!/bin/bash
# session 27: shader clock while k_rof_window runs (GRBM_GUI_ACTIVE / duration): does a lightly loaded GPU clock down?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02y; mkdir -p $O; rm -rf $O/clk
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk -- python3 tools/bench_tvl1occ.py --size 640x480 --cpu none > $O/clk.log 2>&1 || { tail -3 $O/clk.log; exit 1; }
python3 - $O/clk <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, ns = dur[r["Dispatch_Id"]]
    a = acc[name.split("(")[0][:40]]
    a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-42s launches %6d  avg %8.2f us  GRBM_GUI_ACTIVE/8/duration = %.3f GHz" % (k, n, ns / n / 1e3, cyc / 8 / ns))
PY
find $O -name "*.csv" -delete
