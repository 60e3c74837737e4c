#!/usr/bin/env python3
"""gpurun_out/final/ -> profiles/ (tracked): bench line, rocprofv3 kernel stats, PMC traffic JSON.
usage: tools/summarize_profiles.py <tag>      e.g. r01_final"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("OFX_PROF_SRC") or os.path.join(ROOT, "gpurun_out", "final")
DST = os.environ.get("OFX_PROF_DST") or os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"

shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, tag + "_bench.json"))
for extra in ("bench_4k", "bench_f32", "bench_4k_f32"):
    if os.path.exists(os.path.join(SRC, extra + ".json")) and os.path.getsize(os.path.join(SRC, extra + ".json")):
        shutil.copy(os.path.join(SRC, extra + ".json"), os.path.join(DST, tag + "_" + extra + ".json"))
bench = json.load(open(os.path.join(SRC, "bench.json")))

# ---- kernel stats of the bench command ----
def newest(pattern):
    """gpurun merges every collection into the same directory: take the files of the latest one"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = list(csv.DictReader(open(newest(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))[0])))
trace = list(csv.DictReader(open(newest(os.path.join(SRC, "trace", "*", "*_kernel_trace.csv"))[0])))
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 2 --no-cpu --no-4k --no-sor --streams 1 --lockstep 1   (MI355X, f64, 1920x1080 P1)",
       "# the run contains the reference-semantics steps (eps=0.01) AND the fixed-work passes (300 iterations / warp)",
       "# kernel | calls | total ms | average us | % of GPU time", ""]
for r in stats:
    out.append("%-96s %8s %11.3f %10.3f %8.3f" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in trace:
    if "k_tvl1_iter" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = "iter2" if "iter2" in r["Kernel_Name"] else "iter1"
        key = (name, int(r["Grid_Size_X"]))
        agg[key][0] += 1
        agg[key][1] += d
out += ["", "# TV-L1 iteration kernels by grid size (= pyramid level); iter2 = two fused iterations per launch.",
        "# Small averages at a level are dominated by no-op launches behind the stopping iteration.",
        "# kernel grid_threads launches avg_us total_ms"]
full = None
for k, v in sorted(agg.items()):
    out.append("%-6s %9d %7d %9.2f %10.2f" % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3))
# the full-resolution fused launches that did real work (fixed-work pass): compare with bench's HIP-event number.
# Two pyramid levels can share a grid size (1080p at 12 rows/strip and 960x540 at 3 rows/strip are both 184320
# threads), so the launches of the largest grid are split into two duration clusters and the slower one is taken.
gmax = max(k[1] for k in agg if k[0] == "iter2")
durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace
              if "k_tvl1_iter2" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == gmax)
durs = [d for d in durs if d > 5.0]                      # drop no-op launches
lo, hi = durs[0], durs[-1]
for _ in range(20):                                      # 1-D 2-means
    a = [d for d in durs if abs(d - lo) <= abs(d - hi)]
    b = [d for d in durs if abs(d - lo) > abs(d - hi)]
    if not a or not b:
        break
    lo, hi = sum(a) / len(a), sum(b) / len(b)
work = b if b else durs
rocprof_avg = sum(work) / len(work)
out += ["", "# full-resolution k_tvl1_iter2 launches doing real work: %d, rocprofv3 average %.2f us per launch" % (len(work), rocprof_avg),
        "# bench.py (HIP events on the library stream, includes launch gaps): %.2f us per launch" % bench["roofline"]["avg_launch_us"]]
open(os.path.join(DST, tag + "_kernel_stats.txt"), "w").write("\n".join(out) + "\n")

# ---- PMC traffic ----
pm = {"note": "rocprofv3 --pmc passes on tools/pmc_iter.py (40 fixed iterations = 20 launches of k_tvl1_iter2<double>). "
              "FETCH_SIZE is doubled (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md HBM); "
              "WRITE_SIZE as reported. Units: bytes per launch (2 iterations)."}
for sz in ("1920x1080", "3840x2160"):
    vals = {}
    for kind in ("fetch", "write", "sq"):
        f = newest(os.path.join(SRC, "pmc_%s_%s" % (kind, sz), "*", "*_counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "k_tvl1_iter2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
        kt = newest(os.path.join(SRC, "pmc_%s_%s" % (kind, sz), "*", "*_kernel_trace.csv"))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "k_tvl1_iter2" in r["Kernel_Name"]]
        vals["launch_us_" + kind] = sum(d) / len(d)
    nx, ny = map(int, sz.split("x"))
    rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    pm["bytes_per_launch_f64_" + sz] = rd + wr
    pm["detail_" + sz] = {"read_bytes": rd, "write_bytes": wr, "algorithmic_bytes_per_launch": 2 * 120.0 * nx * ny,
                          "traffic_over_algorithmic": (rd + wr) / (2 * 120.0 * nx * ny),
                          "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
                          "valu_active_fraction": 4 * vals["SQ_ACTIVE_INST_VALU"] / 1024 / (vals["launch_us_sq"] * 1e-6 * vals["GRBM_GUI_ACTIVE"] / 8 / (vals["launch_us_fetch"] * 1e-6)),
                          "clock_ghz": vals["GRBM_GUI_ACTIVE"] / 8 / (vals["launch_us_fetch"] * 1e-6) / 1e9,
                          "launch_us": vals["launch_us_fetch"], "valu_insts_per_launch": vals["SQ_INSTS_VALU"],
                          "waves": vals["SQ_WAVES"]}
# ---- the same counters on lockstep-group launches (tools/pmc_group.py) ----
for sz, G in (("1920x1080", 16), ("1920x1080", 5), ("3840x2160", 4)):
    vals = {}
    dsz = sz + ("_g5" if G == 5 else "")
    for kind in ("fetch", "write", "sq"):
        f = newest(os.path.join(SRC, "pmcg_%s_%s" % (kind, dsz), "*", "*_counter_collection.csv"))
        kt = newest(os.path.join(SRC, "pmcg_%s_%s" % (kind, dsz), "*", "*_kernel_trace.csv"))
        if not f or not kt:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "k_tvl1_iter2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "k_tvl1_iter2" in r["Kernel_Name"]]
        vals["launch_us_" + kind] = sum(d) / len(d)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        continue
    nx, ny = map(int, sz.split("x"))
    rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    pm["bytes_per_launch_group%d_f64_%s" % (G, sz)] = rd + wr
    det = {"pairs_per_launch": G, "read_bytes": rd, "write_bytes": wr, "fused_compulsory_bytes_per_launch": G * 120.0 * nx * ny,
           "traffic_over_fused_compulsory": (rd + wr) / (G * 120.0 * nx * ny), "launch_us": vals["launch_us_fetch"],
           "counter_tb_per_s": (rd + wr) / (vals["launch_us_fetch"] * 1e-6) / 1e12}
    if "TCC_HIT_sum" in vals:
        det["l2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        clk = vals["GRBM_GUI_ACTIVE"] / 8 / (vals["launch_us_fetch"] * 1e-6)
        det["clock_ghz"] = clk / 1e9
        det["valu_active_fraction"] = 4 * vals["SQ_ACTIVE_INST_VALU"] / 1024 / (vals["launch_us_sq"] * 1e-6 * clk)
    pm["detail_group%d_%s" % (G, sz)] = det
lines = []
for d_, G in (("trace_group", 16), ("trace_group5", 5)):
    gt = newest(os.path.join(SRC, d_, "*", "*_kernel_stats.csv"))
    if not gt:
        continue
    rows = [r for r in csv.DictReader(open(gt[0])) if "k_tvl1_iter2" in r["Name"]]
    lines.append("# rocprofv3 --kernel-trace --stats -- python3 tools/pmc_group.py 1920x1080 G=%d   (150 launches of %d pairs, fixed work)\n"
                 "# kernel | calls | total ms | average us\n" % (G, G) +
                 "".join("%-100s %6s %10.3f %10.3f\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3) for r in rows))
if lines:
    open(os.path.join(DST, tag + "_group_launch_kernel_stats.txt"), "w").write(
        "".join(lines) + "# bench.py roofline (HIP events, launches of %d pairs): %.2f us per launch\n" % (bench["roofline"].get("pairs_per_launch", 1), bench["roofline"]["avg_launch_us"]))

for sz, ceil in (("1920x1080", "arithmetic alone 92 % of the production time, memory traffic alone 61 % -> VALU-bound (working set inside the 256 MiB Infinity Cache)"),
                 ("3840x2160", "arithmetic alone 81 % of the production time, memory traffic alone 70 % -> neither hides the other (4 waves per SIMD at 128 VGPRs)")):
    if "detail_" + sz not in pm:
        continue
    d, tr = pm["detail_" + sz], pm["bytes_per_launch_f64_" + sz]
    rate = tr / (d["launch_us"] * 1e-6) / 1e12
    pm["limiter_f64_" + sz] = ("co-limited by FP64 issue and memory: VALU active %.0f %% of the launch, counter traffic %.0f MB per launch = %.2f TB/s = %.0f %% "
                               "of the 8 TB/s peak (at the launch time of the counter run, %.0f us, clock %.2f GHz); ceiling variants of the kernel "
                               "(profiles/r02_a_iter2_ceilings_alu_mem.txt, measured once in round 2): " % (100 * d["valu_active_fraction"], tr / 1e6, rate,
                                                                                                      100 * rate / 8, d["launch_us"], d["clock_ghz"])) + ceil
json.dump(pm, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
# the committed bench line carries the traffic measured in THIS collection (bench.py reads the previous file)
gpl = bench["roofline"].get("pairs_per_launch", 1)
key = "bytes_per_launch_%s%s_%s" % ("group%d_" % gpl if gpl > 1 else "", bench["dtype"], bench["roofline"]["kernel"].split("@ ")[1].split(" ")[0])
if key in pm:
    bench["roofline"]["traffic"] = pm[key]
    bench["roofline"]["hbm_frac_counter"] = round(pm[key] / (bench["roofline"]["avg_launch_us"] * 1e-6) / 8e12, 4)
    json.dump(bench, open(os.path.join(DST, tag + "_bench.json"), "w"))
print(json.dumps(pm, indent=1))
print("\n".join(out[-8:]))
