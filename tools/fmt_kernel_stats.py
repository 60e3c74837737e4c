#!/usr/bin/env python3
"""tools/fmt_kernel_stats.py <rocprofv3 output dir> [rows]: the *_kernel_stats.csv of a rocprofv3 --kernel-trace --stats run as a
fixed-width table (kernel names cut to their template arguments, every numeric column kept)."""
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("%-64s %8s %14s %12s %7s %10s %10s" % ("kernel", "calls", "total_ns", "average_ns", "share%", "min_ns", "max_ns"))
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r"^void ", "", r["Name"])
    name = name.split("(")[0] if "(" in name else name
    print("%-64s %8s %14s %12.1f %7s %10s %10s" % (name[:64], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
