#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per kernel and per (kernel, grid) launch count, total, median, p90 of the
LAST `1/reps` of the run.  Usage: rocpd_summary.py results.db [reps] [filter]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
filt = sys.argv[3] if len(sys.argv) > 3 else ""
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from %s d join %s k on d.kernel_id = k.id order by d.start" % (kd, ks)).fetchall()
n = len(rows) // reps
last = rows[-n:]
span = (last[-1][2] - last[0][1]) / 1e3
tot = collections.defaultdict(list)
byg = collections.defaultdict(list)
for name, s, e, gx, gy, wx in last:
    nm = name.split("(")[0]
    nm = nm[:70]
    tot[nm].append((e - s) / 1e3)
    if filt and filt in name:
        byg[(nm[:40], gx // wx, gy)].append((e - s) / 1e3)
busy = sum(sum(v) for v in tot.values())
print("launches %d  span %.1f us  busy %.1f us (%.1f %%)" % (n, span, busy, 100 * busy / span))
for nm, v in sorted(tot.items(), key=lambda x: -sum(x[1]))[:14]:
    v.sort()
    print("  %-70s n=%5d total %9.1f us  median %7.2f  p90 %7.2f" % (nm, len(v), sum(v), v[len(v) // 2], v[int(len(v) * 0.9)]))
for key, v in sorted(byg.items(), key=lambda x: x[0][1]):
    v.sort()
    print("  %-40s wgs %6d x %2d: n=%5d total %9.1f us  median %7.2f  p90 %7.2f  min %.2f" % (key[0], key[1], key[2], len(v), sum(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[0]))
