#!/usr/bin/env python3
"""tools/pmc_summary_r03.py <dir of tools/pmc_round3.sh>: per (size, group, mode) the HBM-side bytes per launch of k_tvl1_iter2 / k_tvl1_iter3
(FETCH_SIZE doubled -- gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md "HBM" -- WRITE_SIZE as
reported, both in KiB), the rate at the launch time of the counter pass, VALU activity, and the rocprofv3 --stats average of
the same launches.  Prints one JSON object (committed as profiles/r03_pmc_group_launches.json)."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]


def newest(pat):
    f = sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


import hashlib
_h = hashlib.sha256()
for _f in ("ofx_tvl1.hip", "ofx_device.h", "ofx_loop.h"):      # what bench.py's kernel_source_sha16() hashes
    _h.update(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optical-flow-1_amd", "csrc", _f), "rb").read())
out = {"note": __doc__.split("Prints")[0].strip(), "kernel_source_sha16": _h.hexdigest()[:16]}
for sz, G in (("1920x1080", 5), ("3840x2160", 4)):
    nx, ny = map(int, sz.split("x"))
    for m, mode, kern in ((0, "strict", "k_tvl1_iter2"), (1, "tolerance", "k_tvl1_iter2"), (1, "tolerance", "k_tvl1_iter3")):
        tag = "%s_g%d_m%d%s" % (sz, G, m, "i3" if kern.endswith("3") else "")
        vals = {}
        for kind in ("fetch", "write", "sq"):
            cc = newest(os.path.join(src, "%s_%s" % (kind, tag), "**", "*_counter_collection.csv"))
            kt = newest(os.path.join(src, "%s_%s" % (kind, tag), "**", "*_kernel_trace.csv"))
            if not cc or not kt:
                continue
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(cc)):
                if kern in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if kern in r["Kernel_Name"]]
            vals["launch_us_" + kind] = sum(d) / len(d)
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            continue
        rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
        comp = G * 120.0 * nx * ny
        rec = {"pairs_per_launch": G, "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": rd + wr,
               "fused_compulsory_bytes_per_launch": comp, "traffic_over_fused_compulsory": (rd + wr) / comp,
               "launch_us_counter_pass": vals["launch_us_fetch"], "counter_tb_per_s": (rd + wr) / (vals["launch_us_fetch"] * 1e-6) / 1e12}
        if "TCC_HIT_sum" in vals:
            rec["l2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
        if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
            clk = vals["GRBM_GUI_ACTIVE"] / 8 / (vals["launch_us_fetch"] * 1e-6)
            rec["clock_ghz"] = clk / 1e9
            rec["valu_active_fraction"] = 4 * vals["SQ_ACTIVE_INST_VALU"] / 1024 / (vals["launch_us_sq"] * 1e-6 * clk)
            rec["valu_insts_per_launch"] = vals["SQ_INSTS_VALU"]
            rec["waves"] = vals["SQ_WAVES"]
            if "SQ_WAVE_CYCLES" in vals:
                rec["wait_any_over_wave_cycles"] = vals.get("SQ_WAIT_ANY", 0.0) / vals["SQ_WAVE_CYCLES"]
                rec["wait_inst_over_wave_cycles"] = vals.get("SQ_WAIT_INST_ANY", 0.0) / vals["SQ_WAVE_CYCLES"]
        st = newest(os.path.join(src, "stats_" + tag, "**", "*_kernel_stats.csv"))
        if st:
            for r in csv.DictReader(open(st)):
                if kern in r["Name"]:
                    rec["rocprofv3_stats_avg_us"] = float(r["AverageNs"]) / 1e3
                    rec["rocprofv3_stats_calls"] = int(r["Calls"])
                    rec["achieved_gbs_fused_compulsory"] = comp / (float(r["AverageNs"]) * 1e-9) / 1e9
                    rec["frac_of_8tbs"] = rec["achieved_gbs_fused_compulsory"] / 8000.0
                    break
        rec["kernel"] = kern
        rec["iterations_per_launch"] = 3 if kern.endswith("3") else 2
        out["%s_group%d_%s%s" % (sz, G, mode, "_iter3" if kern.endswith("3") else "")] = rec
print(json.dumps(out, indent=1))
