#!/bin/bash
# tools/ab_kernel.sh <grep pattern> name=lib.so ... : per-kernel times (rocprofv3 kernel trace of a serial bench run,
# one pair in flight) for several builds of the library
pat=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  rm -rf /tmp/abk
  OFX_LIB_PATH=$R/$path timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/abk -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu --fixed-steps 0 --streams 1 --lockstep 1 > /tmp/abk.json 2> /tmp/abk.err
  echo "== $name: $(python3 -c "import json;print(json.load(open('/tmp/abk.json'))['value'])")"
  python3 $R/tools/trace_busy.py /tmp/abk 0.2 | grep -E "$pat"
done
