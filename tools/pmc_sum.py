#!/usr/bin/env python3
"""tools/pmc_sum.py <rocprofv3 output dir> <kernel-name substring> [...]: totals of every collected counter over the launches
of the kernels whose name contains the substring, the number of those launches and their summed duration (from the
kernel trace of the same run), and the derived figures that only need those: mean resident waves (SQ_WAVE_CYCLES counts
quad-cycles), mean wave lifetime, the split of the wave cycles into parked / issue-stalled / issuing."""
import collections, csv, glob, json, os, sys
src, pats = sys.argv[1], sys.argv[2:]
CLK = 2.4e9


def newest(pat):
    f = sorted(glob.glob(os.path.join(src, "**", pat), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


cc, kt = newest("*_counter_collection.csv"), newest("*_kernel_trace.csv")
for pat in pats:
    tot, n = collections.defaultdict(float), collections.Counter()
    if cc:
        for r in csv.DictReader(open(cc)):
            if pat in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
    secs, launches, t_first, t_last = 0.0, 0, None, None
    if kt:
        for r in csv.DictReader(open(kt)):
            if pat in r["Kernel_Name"]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                secs += (e - s) * 1e-9
                launches += 1
                t_first = s if t_first is None else min(t_first, s)
                t_last = e if t_last is None else max(t_last, e)
    rec = {"kernel": pat, "launches": launches, "kernel_seconds": round(secs, 5),
           "avg_launch_us": round(secs / max(launches, 1) * 1e6, 2),
           "first_to_last_seconds": round((t_last - t_first) * 1e-9, 5) if launches else None,
           "totals": {k: v for k, v in sorted(tot.items())}}
    if "SQ_WAVE_CYCLES" in tot and secs > 0:
        wc = tot["SQ_WAVE_CYCLES"] * 4
        rec["mean_resident_waves"] = round(wc / (secs * CLK), 1)
        if "SQ_WAVES" in tot:
            rec["mean_wave_lifetime_us"] = round(wc / tot["SQ_WAVES"] / CLK * 1e6, 2)
            rec["waves_per_launch"] = round(tot["SQ_WAVES"] / max(launches, 1), 1)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if k in tot:
                rec[k.lower() + "_over_wave_cycles"] = round(tot[k] / tot["SQ_WAVE_CYCLES"], 4)
    if "SQ_INSTS_VALU" in tot and "SQ_WAVES" in tot:
        rec["valu_insts_per_wave"] = round(tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"], 1)
    if "SQ_LDS_BANK_CONFLICT" in tot and "SQ_LDS_IDX_ACTIVE" in tot and tot["SQ_LDS_IDX_ACTIVE"] > 0:
        rec["lds_bank_conflict_fraction"] = round(tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in tot:
        rec["read_gb (FETCH_SIZE doubled, gfx950)"] = round(2 * tot["FETCH_SIZE"] * 1024 / 1e9, 3)
    if "WRITE_SIZE" in tot:
        rec["write_gb"] = round(tot["WRITE_SIZE"] * 1024 / 1e9, 3)
    if "TCC_HIT_sum" in tot:
        rec["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot.get("TCC_MISS_sum", 0.0), 1.0), 4)
    print(json.dumps(rec))
