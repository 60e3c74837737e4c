#!/bin/bash
# session 20: why do group launches write 1.38x their compulsory bytes?  WRITE_SIZE / FETCH_SIZE for G and nt_stores variants
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02t; mkdir -p $O
for spec in "1920x1080 G=1 nt=1" "1920x1080 G=2 nt=0" "1920x1080 G=2 nt=2" "1920x1080 G=16 nt=2" "3840x2160 G=2 nt=2"; do
  tag=$(echo $spec | tr ' =' '__'); rm -rf $O/$tag
  timeout -k 10 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$tag -- python3 tools/pmc_group.py $spec > $O/$tag.log 2>&1 || { echo "$spec failed"; tail -3 $O/$tag.log; exit 1; }
  python3 - "$O/$tag" "$spec" <<'PY'
import csv, glob, sys, collections
d, spec = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_tvl1_iter2" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_tvl1_iter2" in r["Kernel_Name"]]
nx, ny = map(int, spec.split()[0].split("x")); G = int(spec.split()[1][2:])
comp = 120.0 * nx * ny * G
wr = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024; rd = 0.0
print("%-24s launches %4d  avg %8.1f us  write %8.1f MB = %.3f x compulsory   read %8.1f MB = %.3f x   kernel %s" % (
    spec, len(dur), sum(dur) / len(dur), wr / 1e6, wr / (0.4 * comp), rd / 1e6, rd / (0.6 * comp), "nt" if "true" in open(kt).read().split("k_tvl1_iter2")[1][:20] else "plain"))
PY
  find $O/$tag -name "*.csv" -delete
done
