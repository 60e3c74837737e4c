#!/bin/bash
# session 25: counters of the group launches after the store_a guard
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02x; mkdir -p $O
for spec in "1920x1080 G=16" "1920x1080 G=5" "3840x2160 G=4"; do
  tag=$(echo $spec | tr ' =' '__')
  for kind in "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
    k=$(echo $kind | cut -d_ -f1); rm -rf $O/${tag}_$k
    timeout -k 10 120 rocprofv3 --pmc $kind --kernel-trace --output-format csv -d $O/${tag}_$k -- python3 tools/pmc_group.py $spec > $O/${tag}_$k.log 2>&1 || { echo "$spec $kind failed"; exit 1; }
  done
  python3 - "$O" "$tag" "$spec" <<'PY'
import csv, glob, sys, collections, json
O, tag, spec = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for k in ("WRITE", "FETCH"):
    d = "%s/%s_%s" % (O, tag, k)
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tvl1_iter2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        vals[c] = sum(v) / len(v)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_tvl1_iter2" in r["Kernel_Name"]]
    vals["us_" + k] = sum(dur) / len(dur)
nx, ny = map(int, spec.split()[0].split("x")); G = int(spec.split()[1][2:])
rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
comp = 120.0 * nx * ny * G
print(json.dumps({"spec": spec, "read_bytes": rd, "write_bytes": wr, "fused_compulsory": comp, "traffic_over_fused_compulsory": (rd + wr) / comp,
                  "write_over_compulsory": wr / (0.4 * comp), "read_over_compulsory": rd / (0.6 * comp), "launch_us": vals["us_FETCH"],
                  "counter_tb_per_s": (rd + wr) / (vals["us_FETCH"] * 1e-6) / 1e12,
                  "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])}))
PY
  find $O -name "*.csv" -delete
done
