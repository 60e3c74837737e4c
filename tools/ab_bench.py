#!/usr/bin/env python3
"""tools/ab_bench.py name=lib.so[,opt=val...][,env:NAME=VALUE...] ... [--rounds N] [--args "..."]: A/B of library builds / options ON ONE BOX.
Boxes differ by several per cent (clocks, silicon), so variants are only comparable inside one session: every round runs
bench.py once per variant, in turn; prints value / fixed-work / roofline launch time / single-pair ms per run and the medians."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, extra, specs = 3, "--no-cpu --no-sor --no-occ --no-4k", []
it = iter(sys.argv[1:])
for a in it:
    if a == "--rounds":
        rounds = int(next(it))
    elif a == "--args":
        extra = next(it)
    else:
        specs.append(a)
res = {s.split("=")[0]: [] for s in specs}
for r in range(rounds):
    for s in specs:
        name, rest = s.split("=", 1)
        parts = rest.split(",")
        env = dict(os.environ)
        if parts[0]:
            env["OFX_LIB_PATH"] = os.path.join(ROOT, parts[0])
        opts = []
        for o in parts[1:]:
            if o.startswith("env:"):                       # env:NAME=VALUE -- an environment knob of the library
                k, v = o[4:].split("=", 1)
                env[k] = v.replace(";", ",")
            else:
                opts += ["--opt", o]
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"] + extra.split() + opts
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        if p.returncode:
            print(name, "FAILED", p.stderr[-500:])
            continue
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        rec = {"value": d["value"], "fixed": d.get("fixed_work", {}).get("value"), "launch_us": d.get("roofline", {}).get("avg_launch_us"),
               "single_ms": d.get("single_pair", {}).get("device_resident", {}).get("ms_per_pair"),
               "host_ms": d.get("single_pair", {}).get("host_entry", {}).get("ms_per_pair"),
               "fixed_single": d.get("fixed_work", {}).get("single_pair", {}).get("value"),
               "odd_stops": d.get("loop_ends", {}).get("odd_stops"), "strict": d.get("strict", {}).get("value"),
               "launch2_us": d.get("roofline", {}).get("two_iterations_per_launch", {}).get("avg_launch_us")}
        res[name].append(rec)
        print(r, name, json.dumps(rec), flush=True)
for name, v in res.items():
    if v:
        print("MEDIAN", name, {k: statistics.median(x[k] for x in v if x[k] is not None) for k in v[0] if v[0][k] is not None})
