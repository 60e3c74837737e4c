#!/usr/bin/env python3
"""tools/isa_loops.py <file.s> <kernel-name substring> [min instrs]: instruction mix of every loop (label .. backward branch) of
a kernel in device assembly (hipcc --cuda-device-only -S): f64 arithmetic, register moves, DPP, selects, scalar, memory."""
import collections, re, sys
s = open(sys.argv[1]).read()
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if sys.argv[2] in n]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for name in names[:1]:
    a = s.index('\n' + name + ':')
    body = s[a:s.index('.end_amdhsa_kernel', a)].split('\n')
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        lines = [x for x in body[labels[m.group(1)]:i + 1] if re.match(r'\s+[a-z]', x)]
        c = collections.Counter()
        for x in lines:
            op = x.split()[0]
            if 'dpp' in x: c['dpp'] += 1
            elif op.startswith(('v_mov', 'v_accvgpr')): c['v_mov'] += 1
            elif op.startswith('v_') and '_f64' in op: c['f64'] += 1
            elif op.startswith('v_cndmask'): c['cndmask'] += 1
            elif op.startswith('v_'): c['v_other'] += 1
            elif op.startswith('s_'): c['salu'] += 1
            elif op.startswith(('global', 'buffer', 'scratch', 'flat')): c['mem'] += 1
            else: c['other'] += 1
        if len(lines) >= lim:
            print(name[:40], 'loop', labels[m.group(1)], i, 'instrs', len(lines), dict(c))
