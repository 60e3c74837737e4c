#!/bin/bash
# VALU / memory instruction counts of the kernels in one .hip file (device-only assembly for gfx950).
# usage: tools/isa_stats.sh ofx_tvl1.hip [extra hipcc flags] ; prints per kernel: total, v_* , f64 arithmetic, rcp/rsq/sqrt, vgprs
R=/root/repo
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I$R/include -I$R/optical-flow-1_amd/csrc \
  --cuda-device-only -S -o /tmp/isa_$$.s "$@" $R/optical-flow-1_amd/csrc/$F || exit 1
python3 - /tmp/isa_$$.s <<'PY'
import re, sys, collections
cnt = collections.OrderedDict()
name = None
meta_name = None
vg = {}
scr = {}
for ln in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        name = m.group(1)
        cnt[name] = collections.Counter()
        continue
    m = re.match(r'^\s+\.name:\s+(\S+)', ln)
    if m:
        meta_name = m.group(1)
    m = re.match(r'^\s+\.vgpr_count:\s+(\d+)', ln)
    if m and meta_name:
        vg[meta_name] = int(m.group(1))
    m = re.match(r'^\s+\.private_segment_fixed_size:\s+(\d+)', ln)
    if m and meta_name:
        scr[meta_name] = int(m.group(1))
    m = re.match(r'^\s+([a-z_0-9]+)\s', ln)
    if not m or name is None:
        continue
    op = m.group(1)
    if not re.match(r'^(v_|s_|global_|buffer_|ds_|flat_)', op):
        continue
    c = cnt[name]
    c['total'] += 1
    if op.startswith('v_'): c['valu'] += 1
    if '_f64' in op: c['f64'] += 1
    if re.match(r'v_(rcp|rsq|sqrt)_f64', op): c['trans'] += 1
    if re.match(r'v_div_(scale|fmas|fixup)_f64', op): c['div_aux'] += 1
    if re.match(r'(global|buffer)_load', op): c['loads'] += 1
    if re.match(r'(global|buffer)_store', op): c['stores'] += 1
    if op.startswith('ds_'): c['lds'] += 1
    if op.startswith('scratch_'): c['scratch'] += 1
for n, c in cnt.items():
    if not c['total']:
        continue
    print("%-64s vgpr %3s scratch %4s total %5d valu %5d f64 %5d trans %3d div_aux %3d loads %3d stores %3d lds %3d" % (
        n[:64], vg.get(n, '?'), scr.get(n, '?'), c['total'], c['valu'], c['f64'], c['trans'], c['div_aux'], c['loads'], c['stores'], c['lds']))
PY
rm -f /tmp/isa_$$.s
