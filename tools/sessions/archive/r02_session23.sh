#!/bin/bash
# session 23: 32-byte vs 64-byte write requests of k_tvl1_iter2 by pairs per launch (why do group launches write more?)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02w; mkdir -p $O
for spec in "1920x1080 G=1" "1920x1080 G=16" "1920x1080 G=16 nt=2"; do
  tag=$(echo $spec | tr ' =' '__'); rm -rf $O/$tag
  timeout -k 10 90 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum --kernel-trace --output-format csv -d $O/$tag -- python3 tools/pmc_group.py $spec > $O/$tag.log 2>&1 || { echo "$spec failed"; grep -i "error\|invalid\|not" $O/$tag.log | head -5; exit 1; }
  python3 - "$O/$tag" "$spec" <<'PY'
import csv, glob, sys, collections
d, spec = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_tvl1_iter2" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(spec, {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
  find $O/$tag -name "*.csv" -delete
done
