#!/usr/bin/env python3
"""tools/pmc_sor_summary.py <dir of tools/sessions/r04_06_sor_traffic.sh>: HBM-side traffic of the SOR sweep kernels per pixel-sweep.
Per run (solver, mode): sum over the sweep kernel's dispatches of FETCH_SIZE x 2 (gfx950 tallies a wide coalesced read at half its
bytes, MI355X_MICROARCH.md "HBM"; the kernels read 16-byte elements of contiguous runs) and of WRITE_SIZE, both KiB, from separate
passes; divided by the pixel-sweeps the run printed; against SURVEY 8(d)'s unit (56 B Horn-Schunck / 80 B Brox per pixel-sweep).
Prints one JSON object (committed as profiles/r04_sor_traffic.json)."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
import hashlib
_h = hashlib.sha256()
for _f in ("ofx_sor.hip", "ofx_sor_tile.hip", "ofx_device.h", "ofx_loop.h"):      # what bench.py's sor_source_sha16() hashes
    _h.update(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optical-flow-1_amd", "csrc", _f), "rb").read())
out = {"note": __doc__.split("Prints")[0].strip(), "kernel_source_sha16": _h.hexdigest()[:16]}
for log in sorted(glob.glob(os.path.join(src, "fetch_*.log"))):
    tag = os.path.basename(log)[len("fetch_"):-len(".log")]
    meta = None
    for line in open(log):
        if line.startswith("{") and "pixel_sweeps" in line:
            meta = json.loads(line)
    if not meta:
        continue
    unit = 56.0 if meta["which"] == "hs" else 80.0
    rec = {"solver": meta["which"], "group": meta["G"], "size": "%dx%d" % (meta["nx"], meta["ny"]), "options": meta["options"],
           "sweeps_per_pair": meta["sweeps_per_pair"][0], "pixel_sweeps": meta["pixel_sweeps"], "kernels": {}}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    dur = collections.defaultdict(float)
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for cc in glob.glob(os.path.join(src, "%s_%s" % (kind, tag), "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(cc)):
                name = r["Kernel_Name"].split("(")[0].split("<")[0]
                if r["Counter_Name"] == counter:
                    per[name][counter] += float(r["Counter_Value"])
                    if kind == "fetch":
                        cnt[name] += 1
        if kind == "fetch":
            for kt in glob.glob(os.path.join(src, "fetch_%s" % tag, "**", "*_kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(kt)):
                    dur[r["Kernel_Name"].split("(")[0].split("<")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for name, v in per.items():
        if not any(k in name for k in ("k_hs_window", "k_brox_window", "k_hs_tile", "k_brox_wave", "k_brox_tile", "k_brox_sor", "k_hs_sor")):
            continue
        rd, wr = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024, v.get("WRITE_SIZE", 0.0) * 1024
        rec["kernels"][name] = {"dispatches": cnt[name], "read_bytes": rd, "write_bytes": wr,
                                "bytes_per_pixel_sweep": (rd + wr) / meta["pixel_sweeps"], "algorithmic_bytes_per_pixel_sweep": unit,
                                "traffic_over_compulsory": (rd + wr) / meta["pixel_sweeps"] / unit,
                                "kernel_us_in_the_counter_pass": dur.get(name, 0.0)}
    out[tag] = rec
print(json.dumps(out, indent=1))
