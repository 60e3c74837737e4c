#!/usr/bin/env python3
"""tools/trace_busy.py <rocprofv3 out dir> [skip_fraction]: GPU busy fraction (union of kernel intervals) and
time per kernel / per launch geometry from a --kernel-trace csv.  skip_fraction drops the leading part of the
trace (warm-up)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# SOR window kernels: one workgroup per sweep, the workgroup size tells the pyramid level -> keyed by it
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
             int(r["Workgroup_Size_X"]) if "_window" in r["Kernel_Name"] else int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))
            for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + skip * (t1 - t0)
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms  busy(union) %.3f ms = %.1f %%  kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(ev)))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, name, gx, gy in ev:
    short = name.split("(")[0].replace("void ", "")[:40]
    key = (short, gx, gy) if "tvl1_iter" in name or "warp" in name or "_window" in name else (short, 0, 0)
    agg[key][0] += 1
    agg[key][1] += e - s
print("%-42s %9s %4s %8s %10s %9s" % ("kernel", "grid_x", "gy", "calls", "total_ms", "avg_us"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-42s %9d %4d %8d %10.3f %9.2f" % (k[0], k[1], k[2], v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
