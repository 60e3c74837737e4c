#!/bin/bash
# session 28: GPU busy fraction inside the timed region of the driver's 20-step command (union of kernel intervals)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02z; mkdir -p $O; rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-4k --no-sor --no-occ --fixed-steps 0 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(glob.glob(O + "/trace/**/*_kernel_trace.csv", recursive=True)[0]))]
rows.sort()
# the timed region = the last 20-pair pass: find it as the final stretch of duration ms_per_step * steps ending at the last k_to_flo
t_end = max(e for s, e, n in rows if "k_to_flo" in n)
span = d["ms_per_step"] * d["steps"] * 1e6
t0 = t_end - span
sel = [(max(s, t0), min(e, t_end), n) for s, e, n in rows if e > t0 and s < t_end]
busy, cur_s, cur_e = 0, None, None
for s, e, n in sorted(sel):
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
import collections
acc = collections.defaultdict(int)
for s, e, n in sel: acc[n] += e - s
print("value", d["value"], "timed region %.2f ms, GPU busy (union of kernels) %.2f ms = %.1f %%" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
tot = sum(acc.values())
for n, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-42s %7.2f ms summed over streams (%.1f %%)" % (n, v / 1e6, 100.0 * v / tot))
PY
find $O -name "*.csv" -delete
