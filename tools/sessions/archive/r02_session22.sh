#!/bin/bash
# session 22: counters + kernel stats of the group launch of the DRIVER's bench command (20 steps -> 5 pairs per launch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02v; mkdir -p $O
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/pmc_group.py 1920x1080 G=5 > $O/fetch.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/write -- python3 tools/pmc_group.py 1920x1080 G=5 > $O/write.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -- python3 tools/pmc_group.py 1920x1080 G=5 > $O/sq.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/pmc_group.py 1920x1080 G=5 > $O/trace.log 2>&1 || exit 1
python3 - $O <<'PY' | tee $O/summary.json
import csv, glob, json, sys, collections
O = sys.argv[1]; vals = {}
for kind in ("fetch", "write", "sq"):
    f = glob.glob("%s/%s/**/*_counter_collection.csv" % (O, kind), recursive=True)[0]
    kt = glob.glob("%s/%s/**/*_kernel_trace.csv" % (O, kind), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tvl1_iter2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_tvl1_iter2" in r["Kernel_Name"]]
    vals["launch_us_" + kind] = sum(d) / len(d)
st = glob.glob("%s/trace/**/*_kernel_stats.csv" % O, recursive=True)[0]
row = [r for r in csv.DictReader(open(st)) if "k_tvl1_iter2" in r["Name"]][0]
G, nx, ny = 5, 1920, 1080
rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
clk = vals["GRBM_GUI_ACTIVE"] / 8 / (vals["launch_us_fetch"] * 1e-6)
print(json.dumps({"bytes_per_launch": rd + wr, "detail": {"pairs_per_launch": G, "read_bytes": rd, "write_bytes": wr,
      "fused_compulsory_bytes_per_launch": G * 120.0 * nx * ny, "traffic_over_fused_compulsory": (rd + wr) / (G * 120.0 * nx * ny),
      "launch_us": vals["launch_us_fetch"], "counter_tb_per_s": (rd + wr) / (vals["launch_us_fetch"] * 1e-6) / 1e12,
      "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), "clock_ghz": clk / 1e9,
      "valu_active_fraction": 4 * vals["SQ_ACTIVE_INST_VALU"] / 1024 / (vals["launch_us_sq"] * 1e-6 * clk)},
      "kernel_stats": {"name": row["Name"][:60], "calls": int(row["Calls"]), "average_us": float(row["AverageNs"]) / 1e3}}))
PY
find $O -name "*.csv" -delete
