#!/usr/bin/env python3
"""tools/trace_budget.py <rocprofv3 out dir> [--all]: time budget of bench.py's TIMED REGION from a --kernel-trace csv.

The region is what lies between the two marker fills bench.py issues with OFX_BENCH_MARK=1 (a 7777-element torch fill
before t0 and after the closing fence).  Several contexts (HIP streams) run concurrently, so kernel durations overlap: the
table gives, per kernel class, calls / summed duration / share of the summed duration, plus per stream the busy time and
the idle gaps between consecutive launches, and the union-busy fraction of the region.  Launches of the iteration kernel
whose duration is below 30 % of the median of their geometry are counted as no-ops (launches behind a stop)."""
import collections
import csv
import glob
import os
import statistics
import sys

d = sys.argv[1]
whole = "--all" in sys.argv
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
skey = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]),
             int(r["Grid_Size_Z"]), r[skey]) for r in rows)
marks = [e for e in ev if "FillFunctor" in e[2]]
print("# trace %s: %d kernels, %d marker fills, stream key %s" % (os.path.basename(f), len(ev), len(marks), skey))
if not whole:
    if len(marks) < 2:
        raise SystemExit("no marker pair in the trace (run bench.py with OFX_BENCH_MARK=1) -- or pass --all")
    lo, hi = marks[0][1], marks[1][0]
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
span = t1 - t0


def union(evs):
    busy, cs, ce = 0, None, None
    for s, e, *_ in sorted(evs):
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + (ce - cs if ce is not None else 0)


tot = sum(e[1] - e[0] for e in ev)
print("region: first kernel start -> last kernel end %.3f ms; union busy %.3f ms = %.1f %%; summed kernel time %.3f ms "
      "(average overlap %.2f); %d launches" % (span / 1e6, union(ev) / 1e6, 100.0 * union(ev) / span, tot / 1e6, tot / span, len(ev)))


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0][:44]


agg = collections.defaultdict(list)
for s, e, name, gx, gy, gz, st in ev:
    k = short(name)
    if "tvl1_iter" in name or "warp" in name:
        k = "%s grid %dx%dx%d" % (k, gx, gy, gz)
    agg[k].append(e - s)
print("\n%-66s %7s %10s %7s %9s %9s" % ("kernel (iteration / warp kernels per launch geometry)", "calls", "sum_ms", "share", "avg_us", "noop"))
cls = collections.defaultdict(float)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    med = statistics.median(v)
    noop = sum(1 for x in v if x < 0.3 * med) if "tvl1_iter" in k else 0
    noop_ms = sum(x for x in v if x < 0.3 * med) / 1e6 if "tvl1_iter" in k else 0.0
    print("%-66s %7d %10.3f %6.1f%% %9.2f %5d (%.3f ms)" % (k, len(v), sum(v) / 1e6, 100.0 * sum(v) / tot, sum(v) / len(v) / 1e3, noop, noop_ms))
    c = ("iteration kernels" if "tvl1_iter" in k else "warp" if "warp" in k else "finalize (poll record)" if "finalize" in k
         else "pyramid / level set-up / output")
    cls[c] += sum(v)
    if noop_ms:
        cls["  of which no-op iteration launches"] += noop_ms * 1e6
print("\nclass shares of the summed kernel time:")
for c, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print("  %-40s %9.3f ms %6.1f %%" % (c, v / 1e6, 100.0 * v / tot))

print("\nper stream: launches, busy (sum of its kernels), idle between its consecutive launches, longest gaps")
bys = collections.defaultdict(list)
for e in ev:
    bys[e[6]].append(e)
for st, v in sorted(bys.items()):
    v.sort()
    gaps = [max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)]
    big = sorted(gaps)[-3:]
    print("  stream %-6s %6d launches  busy %8.3f ms  idle %8.3f ms (%.1f %% of its active span %.3f ms)  gaps > 20 us: %d  max %s us"
          % (st, len(v), sum(e[1] - e[0] for e in v) / 1e6, sum(gaps) / 1e6, 100.0 * sum(gaps) / max(1, v[-1][1] - v[0][0]),
             (v[-1][1] - v[0][0]) / 1e6, sum(1 for g in gaps if g > 20000), [round(g / 1e3, 1) for g in big]))
