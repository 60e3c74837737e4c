#!/usr/bin/env python3
"""bench.py -- TV-L1 multiscale throughput on MI355X, BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workloads (--workload):
  1080p     (default, BASELINE configs[1], the configuration the metric is quoted on)  One "step" = one synthetic
            1920x1080 image pair (P1 of SURVEY.md 8d) through the TV-L1 multiscale solve with the reference's default
            parameters (nscales=5 warps=5 tau=0.25 lambda=0.15 theta=0.3 zfactor=0.5 epsilon=0.01) in f64 storage,
            inputs resident in HBM, result = the .flo payload in HBM.  Every rank solves its own K pairs: weak scaling.
  4k-batch  (BASELINE configs[4])  ONE batch of K (default 64) distinct synthetic 3840x2160 pairs, pair k on rank
            k mod N (8 pairs per GPU at N = 8, all 64 on one GPU at N = 1): strong scaling.
Work is counted exactly like the reference prints it:  work = sum_scales sum_warps n_iter * nx_s * ny_s  [pixel-
iterations], and `value` = whole-job Mpix*warp-iters/s = (work of all ranks over the timed steps) / (max over ranks of
the wall time of those steps, barrier + device sync on both sides) / 1e6.

N > 1: no data-path collective (pairs are independent); the timed region ends with the RCCL gather of the float32 .flo
payloads to rank 0.  The pairs of a rank are solved in ROUNDS and the gather of round r runs (asynchronously, RCCL's
own stream) while round r + 1 computes, so only the last round's transfer is exposed (`gather_ms`).

The line also carries: `fixed_work` (option "fixed_work": every warp runs exactly 300 iterations);
`roofline` / `roofline_4k` for the dominant kernel k_tvl1_iter2 as the job launches it (one launch = a lockstep group of
up to 16 pairs), measured on fixed-work passes with HIP events on the library's own stream, with the one-pair launch beside
it (`single_pair`); `sor` (BASELINE configs 3 / 4: Horn-Schunck and Brox solves, exact and tolerance mode); `occ` (TV-L1 with occlusions); `cpu_baseline` (the compiled
reference, oracle/_ref, on the host cores; rank 0 at N = 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAR = dict(tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5, epsilon=0.01)
ELEM = {"f64": 8.0, "f32": 4.0}
HBM_PEAK_GBS = 8000.0
ARITH = {"strict": "f64 storage and arithmetic, every per-pixel operation in the reference's IEEE order (glibc-exact hypot, IEEE divisions): "
                   "flows and iteration tables bit-identical to the reference (tests/test_gpu_tvl1.py)",
         "tolerance": "f64 storage and arithmetic; in the dual update sqrt(x^2 + y^2) for libm's hypot and one reciprocal per denominator "
                      "(v_rsq_f64 / v_rcp_f64 + one refinement step, ~2^-45 relative), primal division by reciprocal: not bit-identical -- AEPE vs the "
                      "reference ~1e-12 px on every BASELINE config (this line's cpu_baseline.gpu_vs_this_reference_run; bar: 1e-4), iteration "
                      "tables equal on all of them"}
WORKLOADS = {"1080p": (1920, 1080), "4k-batch": (3840, 2160)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="pairs per rank (1080p, default 64) / pairs of the batch (4k-batch, default 64)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the timed K-step region (each one bracketed by barrier + sync); the median one is reported")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1080p")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--mode", choices=["tolerance", "strict"], default="tolerance",
                    help="f64 arithmetic of the headline: 'tolerance' = the dual update without its last-bit fidelity (sqrt(x^2+y^2), "
                         "reciprocals; AEPE vs the reference ~1e-12, north_star's bar 1e-4), 'strict' = bit-identical to the "
                         "reference.  The other mode is measured beside it (object `strict` / `tolerance`)")
    ap.add_argument("--nx", type=int, default=0, help="override the workload's image width (tests / rehearsals)")
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--pair", default="P1")
    ap.add_argument("--fixed-steps", type=int, default=2, help="fixed-work passes for the roofline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-4k", action="store_true", help="skip the roofline_4k leg")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the leg that repeats the timed command in the other f64 mode")
    ap.add_argument("--no-single", action="store_true", help="skip the single_pair leg (one pair at a time, device-resident and host entry)")
    ap.add_argument("--no-sor", action="store_true", help="skip the sor leg (BASELINE configs 3 / 4)")
    ap.add_argument("--no-cli", action="store_true", help="skip the cli leg (end-to-end time of bin/tvl1flow in a fresh process)")
    ap.add_argument("--no-occ", action="store_true", help="skip the occ leg (TV-L1 with occlusions, SURVEY 8f.1)")
    ap.add_argument("--lockstep", type=int, default=0,
                    help="pairs per lockstep group (they share every kernel launch of one context); 0 = the library's choice")
    ap.add_argument("--concurrency", type=int, default=0, help="override the library's concurrency hint (0 = streams)")
    ap.add_argument("--variants", type=int, default=8, help="1080p: distinct synthetic pairs cycled through by the steps")
    ap.add_argument("--streams", type=int, default=4,
                    help="lockstep groups in flight per GPU, each on its own context / HIP stream (SURVEY 8e: >= 2)")
    ap.add_argument("--rounds", type=int, default=0,
                    help="rounds per rank (the gather of a round overlaps the next round's compute); 0 = automatic: "
                         "1 on one GPU, 2 (3/4 + 1/4 of the pairs) on several when a rank has >= 32 pairs")
    ap.add_argument("--check", action="store_true",
                    help="N > 1: rank 0 re-solves every pair of every rank and compares the gathered payloads byte for byte")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value for every context (A/B runs)")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--rows2", type=int, default=0)
    return ap.parse_args()


T_START = time.perf_counter()


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process has not touched the GPU (not even
    imported torch) -- it starts the N ranks as fresh child processes through torch.distributed.run, relays rank 0's
    JSON line (the children inherit stdout) and returns their exit code.  Nothing is re-exec'ed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (a.gpus, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


def split_rounds(n, rounds, nstreams):
    """[(first, count)] slices of a rank's n pairs.  Two rounds: the last one ~1/4 of the pairs, a multiple of the
    number of contexts where possible (every context then gets one lockstep group per round)."""
    rounds = min(rounds, n)
    if rounds <= 1:
        return [(0, n)]
    out, first = [], 0
    for r in range(rounds):
        left = rounds - r
        if left == 1:
            cnt = n - first
        else:
            cnt = (n - first) * (2 ** (left - 1) + 1) // (2 ** left)     # 2 rounds: 3/4 + 1/4
            if cnt >= nstreams:
                cnt -= cnt % nstreams
            cnt = max(1, min(cnt, n - first - (left - 1)))
        out.append((first, cnt))
        first += cnt
    return out


def cpu_baseline(synth, nx, ny, pair, gpu_flows=None):
    """Reference (oracle/_ref) on the host cores: a bounded sample of the 1080p workload (4 pairs through the multiscale
    call) + the inner loop alone.  gpu_flows(k, I0, I1) -> {mode: (u, v)}: the GPU's flows of the same pairs, compared with
    the reference's (AEPE, the parity bar of north_star)."""
    import oracle
    if oracle.have_ref():
        cpu = oracle.Ref()
    else:
        if not os.path.exists(oracle.ORACLE_SO):
            oracle.build()
        cpu = oracle.Oracle()
    port = oracle.Oracle() if os.path.exists(oracle.ORACLE_SO) else None
    cores = min(oracle.host_cores(), 32)
    cpu.set_num_threads(cores)
    log("cpu_baseline: %s on %d threads (affinity %d)" % (cpu.kind, cores, len(os.sched_getaffinity(0))))
    NPAIR = 4
    pairs = [synth.pair(pair, nx, ny, k) for k in range(NPAIR)]
    t0 = time.perf_counter()
    ref_flows = []
    for I0, I1 in pairs:
        r_ = cpu.tvl1_multiscale(I0, I1, **PAR)
        ref_flows.append((r_[0], r_[1]))
    t_ms = time.perf_counter() - t0
    log("cpu_baseline: %d multiscale calls %.2f s" % (NPAIR, t_ms))
    out = {"unit": "Mpix*warp-iters/s", "cores": cores, "kind": cpu.kind,
           "sample": "%d pairs %s %dx%d (batch variants 0..%d), same parameters, Dual_TVL1_optic_flow_multiscale (%.2f s)"
                     % (NPAIR, pair, nx, ny, NPAIR - 1, t_ms), "seconds": round(t_ms, 3),
           "build": "-O3 -fopenmp, generic x86-64"}
    if gpu_flows is not None:
        par = {}
        for k, ((I0, I1), (ur, vr)) in enumerate(zip(pairs, ref_flows)):
            for mode, (ug, vg) in gpu_flows(k, I0, I1).items():
                e = par.setdefault(mode, {"aepe_max": 0.0, "max_abs": 0.0})
                e["aepe_max"] = max(e["aepe_max"], float(np.mean(np.hypot(ug - ur, vg - vr))))
                e["max_abs"] = max(e["max_abs"], float(max(np.abs(ug - ur).max(), np.abs(vg - vr).max())))
        out["gpu_vs_this_reference_run"] = dict(par, pairs=NPAIR, tolerance_aepe=1e-4,
                                                note="flows of the same %d pairs on the GPU against the reference's, per f64 mode" % NPAIR)
    # work of those calls: iteration counts from the port (bit-identical loop, OMP_NUM_THREADS=1 parity-tested)
    if port is not None:
        port.set_num_threads(cores)
        sizes = [(nx, ny)]
        for _ in range(1, PAR["nscales"]):
            sizes.append(cpu.zoom_size(sizes[-1][0], sizes[-1][1], PAR["zfactor"]))
        work = 0.0
        for I0, I1 in pairs:
            _, _, iters, _ = port.tvl1_multiscale(I0, I1, **PAR)
            work += float(sum(int(iters[s].sum()) * sizes[s][0] * sizes[s][1] for s in range(PAR["nscales"])))
        out["value"] = round(work / t_ms / 1e6, 3)
    I0, I1 = pairs[0]
    # fixed-work inner loop: Dual_TVL1_optic_flow, 2 warps, eps=0 -> 300 iterations each at full resolution
    z = np.zeros((ny, nx))
    t0 = time.perf_counter()
    if cpu.kind == "reference":
        cpu.tvl1_single_scale(I0, I1, z, z, tau=PAR["tau"], lam=PAR["lam"], theta=PAR["theta"], warps=2, epsilon=0.0)
        n_it = 600
    else:
        _, _, it, _ = cpu.tvl1_single_scale(I0, I1, z, z, tau=PAR["tau"], lam=PAR["lam"], theta=PAR["theta"], warps=2,
                                            epsilon=0.0)
        n_it = sum(it)
    t_fx = time.perf_counter() - t0
    log("cpu_baseline: fixed-work inner loop %.2f s" % t_fx)
    out["fixed_work"] = {"value": round(n_it * nx * ny / t_fx / 1e6, 3), "unit": "Mpix*iters/s",
                         "sample": "Dual_TVL1_optic_flow %dx%d, 2 warps, eps=0 (%d iterations, %.2f s)" % (nx, ny, n_it, t_fx)}
    return out


def kernel_source_sha16():
    """hash of the sources the TV-L1 iteration kernels are compiled from: stored with the counter passes (tools/pmc_summary_r03.py),
    compared here -- counters collected on other kernel sources are reported as stale, not silently reused"""
    import hashlib
    h = hashlib.sha256()
    for f in ("ofx_tvl1.hip", "ofx_device.h", "ofx_loop.h"):
        h.update(open(os.path.join(ROOT, "optical-flow-1_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def sor_source_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("ofx_sor.hip", "ofx_sor_tile.hip", "ofx_device.h", "ofx_loop.h"):
        h.update(open(os.path.join(ROOT, "optical-flow-1_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def sor_traffic(solver, mode):
    """HBM-side traffic of the sweep kernels per pixel-sweep from the committed counter passes (tools/sessions/r04_06_sor_traffic.sh:
    one full-resolution single-scale group solve per --pmc pass), with the stale flag of roofline.traffic"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r04_sor_traffic.json")))
    except Exception:
        return None
    tags = {("hs", "exact"): ["hs_exact_global", "hs_exact_lds"], ("hs", "tolerance"): ["hs_tolerance_k2", "hs_tolerance_k4"],
            ("brox", "exact"): ["brox_exact_global", "brox_exact_lds"],
            ("brox", "tolerance"): ["brox_tolerance", "brox_redblack_tile_k4", "brox_redblack_two_launches_per_sweep"]}[(solver, mode)]
    out = {"source": "profiles/r04_sor_traffic.json (builder-run rocprofv3 --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md; not "
                     "measured by this run)", "stale": j.get("kernel_source_sha16") != sor_source_sha16(), "kernels": {}}
    for t in tags:
        for name, r in j.get(t, {}).get("kernels", {}).items():
            out["kernels"]["%s (%s)" % (name.replace("void ", ""), t)] = {"bytes_per_pixel_sweep": round(r["bytes_per_pixel_sweep"], 1),
                                                                          "traffic_over_compulsory": round(r["traffic_over_compulsory"], 3)}
    return out


def load_pmc():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}


def roofline_of(ctx, solve_one, precision, nx, ny, passes, group=None, mode="strict"):
    """Fixed-work passes with HIP events (on the library's stream) around the iteration launches of every level: of ONE pair
    alone (per-level table, `single_pair`), and -- group = (G, solve_group) -- of a lockstep group of G pairs, which is how
    the timed region launches the kernel (one launch = G pairs; blockIdx.y/z = pair).  The roofline is quoted on the group
    launch when given: bytes and duration both per launch of G pairs.  Returns (roofline dict for the full-resolution
    k_tvl1_iter2 launches, per-level list, work, seconds).

    Bytes: the kernel fuses TWO iterations per launch and touches every stream once per launch, so its compulsory
    HBM traffic is 15 storage elements per pixel PER LAUNCH (read U,P1,P2,A,R = 9, write U,P1,P2 = 6): `achieved` and
    `frac` use that figure and cannot exceed 1.  SURVEY 8(d)'s accounting unit (15 elements per pixel per ITERATION,
    what an unfused kernel has to move) is kept next to it as `algorithmic_equivalent_*`: it says how fast a
    one-iteration-per-launch kernel would have to stream to keep up, and may exceed the HBM peak."""
    elem = ELEM[precision]
    ctx.set_option("concurrency", 1)
    ctx.set_option("profile", 1)
    ctx.set_option("fixed_work", 1)
    solve_one()                                          # warm
    ctx.synchronize()
    t0 = time.perf_counter()
    work, ns = 0.0, PAR["nscales"]
    lv_ms, lv_n = [0.0] * ns, [0] * ns
    st = None
    for _ in range(passes):
        work += solve_one()
        st = ctx.stats()
        for s_ in range(ns):
            lv_ms[s_] += st.iter_ms[s_]
            lv_n[s_] += st.iter_launches[s_]
    ctx.synchronize()
    secs = time.perf_counter() - t0
    ctx.set_option("profile", 0)
    ctx.set_option("fixed_work", 0)
    us_iter = lv_ms[0] * 1e3 / max(lv_n[0], 1)            # per ITERATION (stats count iterations)
    F1 = max(int(st.fused[0]), 1)                         # iterations per launch of the kernel the library picked (2, or 3: k_tvl1_iter3)
    us_single = F1 * us_iter
    fused = 15.0 * elem * nx * ny                         # bytes per pair and launch, compulsory for the fused kernel
    G, us_launch, n_launch, F = 1, us_single, int(lv_n[0] // F1), F1
    if group:
        G, solve_group = group
        ctx.set_option("profile", 1)
        ctx.set_option("fixed_work", 1)
        solve_group()                                    # warm (level arrays of G pairs)
        g_ms, g_n = 0.0, 0
        for _ in range(max(1, passes)):
            st_g = solve_group()
            F = max(int(st_g[0].fused[0]), 1)
            g_ms += st_g[0].iter_ms[0]                   # the group's launches, recorded on every member
            g_n += st_g[0].iter_launches[0] // F
        ctx.synchronize()
        ctx.set_option("profile", 0)
        ctx.set_option("fixed_work", 0)
        us_launch, n_launch = g_ms * 1e3 / max(g_n, 1), g_n
    ach = G * fused / (us_launch * 1e-6) / 1e9
    equiv = F * ach
    tname = "double" if precision == "f64" else "float"
    roof = {"bound": "hbm", "iterations_per_launch": F,
            "kernel": "k_tvl1_iter%d<%s> @ %dx%d (%d fused iterations per launch, %d pair%s per launch)" % (F, tname, nx, ny, F, G, "" if G == 1 else "s"),
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": None, "avg_launch_us": round(us_launch, 3), "launches": n_launch, "pairs_per_launch": G,
            "fused_algorithmic_bytes_per_launch": G * fused,
            "bytes_model": "15 storage elements/px per pair and LAUNCH (%d fused iterations): read U,P1,P2,A,R + write U,P1,P2" % F,
            "algorithmic_equivalent_bytes_per_launch": float(F) * G * fused,
            "algorithmic_equivalent_gbs": round(equiv, 1),
            "algorithmic_equivalent_frac": round(equiv / HBM_PEAK_GBS, 4),
            "algorithmic_equivalent_note": "SURVEY 8(d): 15 elements/px per ITERATION x %d iterations per launch; the rate an "
                                           "unfused kernel would need -- may exceed the HBM peak, not a bandwidth" % F,
            "mpix_iters_per_s": round(float(F) * G * nx * ny / us_launch, 1), "arithmetic_mode": mode if precision == "f64" else "f32"}
    if F == 3:
        roof["note"] = ("three fused iterations move the compulsory streams once per THREE iterations: `frac` (compulsory bytes per launch / "
                        "launch time / peak) is lower than the two-iteration kernel's although the iteration rate is ~30 % higher -- "
                        "by its ceiling builds the launch follows its memory path (~5 TB/s of real traffic for this stream mix), FP64 issue "
                        "25 % below it (DESIGN 5.1, profiles/r04_ab_iter3_alignment.txt); `two_iterations_per_launch` is the same "
                        "launch shape with option fuse3 = 0")
    if group and F == 3:                                  # the same launch shape with two iterations per launch, for comparison
        ctx.set_option("fuse3", 0)
        ctx.set_option("profile", 1)
        ctx.set_option("fixed_work", 1)
        solve_group()
        st2 = solve_group()
        ctx.synchronize()
        ctx.set_option("profile", 0)
        ctx.set_option("fixed_work", 0)
        ctx.set_option("fuse3", 2)
        us2 = st2[0].iter_ms[0] * 1e3 / max(st2[0].iter_launches[0] // 2, 1)
        a2 = G * fused / (us2 * 1e-6) / 1e9
        roof["two_iterations_per_launch"] = {"kernel": "k_tvl1_iter2<%s>, option fuse3 = 0" % tname, "avg_launch_us": round(us2, 3),
                                             "achieved": round(a2, 1), "frac": round(a2 / HBM_PEAK_GBS, 4),
                                             "mpix_iters_per_s": round(2.0 * G * nx * ny / us2, 1)}
    if group:
        a1 = fused / (us_single * 1e-6) / 1e9
        roof["single_pair"] = {"avg_launch_us": round(us_single, 3), "achieved": round(a1, 1), "frac": round(a1 / HBM_PEAK_GBS, 4),
                               "note": "the same kernel launched for one pair alone: start-up and drain of every launch exposed "
                                       "(at 1080p one pair is 2880 waves, less than one round of the 3072 wave slots)"}
    pmc = load_pmc()
    # counter passes on the group launches, per f64 mode (tools/pmc_round3.sh): preferred when they cover this launch; the newest
    # collection wins, and its source hash says whether it was taken on the kernels this run executes
    r3, pmc_file, pmc_sha = None, None, None
    for cand in ("r04_pmc_group_launches.json", "r03_pmc_group_launches.json"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", cand)))
        except Exception:
            continue
        r3 = j.get("%dx%d_group%d_%s%s" % (nx, ny, G, mode, "_iter3" if F == 3 else ""))
        if r3:
            pmc_file, pmc_sha = cand, j.get("kernel_source_sha16")
            break
    if r3 and precision == "f64":
        roof["traffic"] = r3["bytes_per_launch"]
        roof["traffic_stale"] = pmc_sha != kernel_source_sha16()
        roof["traffic_source"] = ("profiles/%s (builder-run rocprofv3 --pmc passes on the same launch shape and "
                                  "mode, FETCH_SIZE doubled per MI355X_MICROARCH.md; not measured by this run; kernel sources at "
                                  "collection %s, now %s)" % (pmc_file, pmc_sha, kernel_source_sha16()))
        roof["hbm_frac_counter"] = round(r3["bytes_per_launch"] / (us_launch * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        roof["traffic_over_compulsory"] = round(r3["traffic_over_fused_compulsory"], 4)
        roof["valu_active"] = round(r3["valu_active_fraction"], 3)
        roof["limiter"] = ("strict kernel: co-limited by FP64 issue (VALU active 0.70) and memory (counter traffic 5.6-5.7 TB/s at the counter "
                           "pass's launch time)" if mode == "strict" else
                           ("tolerance kernel, three iterations per launch: follows its memory path -- streams at %.1f TB/s by the counters, "
                            "memory ceiling (no arithmetic) 266 us / ALU ceiling (cache-resident loads, no stores) 219 us / production "
                            "291 us on one box at 1080p x 5; VALU active %.2f" % (r3["counter_tb_per_s"], r3["valu_active_fraction"])) if F == 3 else
                           "tolerance kernel: memory-bound -- counter traffic 6.0 (1080p) / 6.3 (4K) TB/s at the counter pass's launch time, "
                           "the rate the guide gives as achievable for HBM3E on this part; VALU active 0.45-0.48")
        levels = [{"size": "%dx%d" % (st.nx[s_], st.ny[s_]), "iter_us": round(lv_ms[s_] * 1e3 / max(lv_n[s_], 1), 2),
                   "ms_per_step": round(lv_ms[s_] / passes, 2)} for s_ in range(ns)]
        return roof, levels, work, secs
    key = "%s_%dx%d" % (precision, nx, ny)
    tr = pmc.get("bytes_per_launch_group%d_%s" % (G, key)) if G > 1 else pmc.get("bytes_per_launch_" + key)
    src = "profiles/pmc_traffic.json (builder-run rocprofv3 --pmc passes, not measured by this run)"
    if not tr and G > 1 and pmc.get("bytes_per_launch_" + key):
        tr = G * pmc["bytes_per_launch_" + key]
        src += "; counter value of a one-pair launch x %d pairs" % G
    if tr:
        det = {}
        if precision == "f64":
            det = (pmc.get("detail_group%d_%dx%d" % (G, nx, ny)) if G > 1 else None) or pmc.get("detail_%dx%d" % (nx, ny), {})
        roof["traffic"] = tr
        roof["traffic_source"] = src
        roof["hbm_frac_counter"] = round(tr / (us_launch * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if "valu_active_fraction" in det and mode == "strict":
            roof["valu_active"] = round(det["valu_active_fraction"], 3)
        if mode != "strict":
            roof["traffic_source"] += "; counter passes ran the strict kernel, which moves the same streams"
    lim = pmc.get("limiter_" + key) or pmc.get("limiter")
    if lim and mode == "strict":
        roof["limiter"] = lim
    elif mode == "tolerance":
        roof["limiter"] = ("tolerance-mode kernel: ~30 % fewer FP64 instructions than the strict one (no glibc-exact hypot, reciprocals "
                           "instead of IEEE quotients), same memory streams -- closer to the memory ceiling of the strict kernel's "
                           "co-limited profile (profiles/r02_a_iter2_ceilings_alu_mem.txt)")
    levels = [{"size": "%dx%d" % (st.nx[s_], st.ny[s_]), "iter_us": round(lv_ms[s_] * 1e3 / max(lv_n[s_], 1), 2),
               "ms_per_step": round(lv_ms[s_] / passes, 2)} for s_ in range(ns)]
    return roof, levels, work, secs


def sor_leg(ofx_mod, synth, local, dev, with_cpu=True):
    """BASELINE configs 3 and 4 (parity-test cases, not the headline): Horn-Schunck 1920x1080 and Brox 1280x720, in both modes
    of the library.  `exact` (the default; also at the top level of each entry, where round 3 had it): the reference's sweep
    order, bit-identical to the reference.  `tolerance` (option sor_exact = 0, csrc/ofx_sor_tile.hip): re-ordered sweeps inside
    north_star's bar (AEPE < 1e-4 against the reference) -- Horn-Schunck four-colour sweeps, K per launch on LDS tiles; Brox a
    checkerboard of tiles swept in the reference's order on the finest level, red-black below -- with the AEPE and max |delta|
    of pair P0 against the reference run of the same pair on this host (one thread: the reference's only deterministic
    configuration) and, beside it, the reference's own 1-vs-8-thread spread on that pair.
    `one_pair`: one solve through the host entry point (the reference's calling convention, host arrays in / out).  `batch`: 48
    device-resident pairs (P0 + 47 P1 variants) through ofx_hs_batch_dev / ofx_brox_batch_dev, lockstep groups of 16 pairs on 3
    contexts; every flow is bit-identical to the pair solved alone (tests/test_gpu_sor.py, tests/test_gpu_sor_tile.py).
    Algorithmic bytes per sweep: 56 B/px (HS), 80 B/px (Brox) -- SURVEY 8(d)."""
    import torch
    out = {}
    solo = ofx_mod.Ofx(local, ofx_mod.F64)
    ctxs = [ofx_mod.Ofx(local, ofx_mod.F64) for _ in range(3)]
    for c in ctxs:
        c.set_option("lockstep", 16)
    NB = 48
    ref = None
    if with_cpu:
        import oracle
        if oracle.have_ref():
            ref = oracle.Ref()
    for name, host_fn, batch_fn, ref_name, size, bpp, kw in (
            ("hs_cfg3", solo.hs_pyramidal, ofx_mod.hs_batch_dev, "hs_pyramidal", (1920, 1080), 56.0,
             dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)),
            ("brox_cfg4", solo.brox_spatial, ofx_mod.brox_batch_dev, "brox_spatial", (1280, 720), 80.0,
             dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15))):
        nx, ny = size

        def rec(work, dt, extra):
            mps = work / dt / 1e6
            r = {"seconds": round(dt, 4), "mpix_sweeps_per_s": round(mps, 1), "algorithmic_gbs": round(mps * bpp / 1e3, 1),
                 "frac_of_hbm_peak": round(mps * bpp / 1e3 / HBM_PEAK_GBS, 5)}
            r.update(extra)
            return r
        I1, I2 = synth.pair("P0", nx, ny)
        ins = [synth.pair_device("P0" if k == 0 else "P1", nx, ny, k, dev) for k in range(NB)]
        flo = torch.empty((NB, ny, nx, 2), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        args = ([t[0].data_ptr() for t in ins], [t[1].data_ptr() for t in ins], [flo[k].data_ptr() for k in range(NB)], nx, ny)
        modes = {}
        flows = {}
        for mode, exact in (("exact", 1), ("tolerance", 0)):
            for c in ctxs + [solo]:
                c.set_option("sor_exact", exact)
            res = host_fn(I1, I2, **kw)                      # warm (arena, clocks, result planes)
            reps = []
            for _ in range(3):
                t0 = time.perf_counter()
                host_fn(I1, I2, out=res, **kw)
                reps.append(time.perf_counter() - t0)
            dt = sorted(reps)[1]                             # median of three (single solves vary by ~10 % from run to run)
            st = solo.stats()
            flows[mode] = (res[0].copy(), res[1].copy())
            one = rec(st.work_pix_iters, dt, {"sweeps": int(st.iterations().sum()), "pair": "P0, host arrays in/out",
                                              "repetitions": [round(r_, 4) for r_ in reps]})
            batch_fn(ctxs, *args, **kw)                      # warm
            t0 = time.perf_counter()
            work = batch_fn(ctxs, *args, **kw)
            dt = time.perf_counter() - t0
            modes[mode] = {"one_pair": one,
                           "batch": rec(sum(work), dt, {"pairs": NB, "contexts": len(ctxs), "lockstep_group": 16,
                                                        "ms_per_pair": round(dt / NB * 1e3, 2),
                                                        "pairs_desc": "P0 + 47 P1 variants, device-resident"})}
        for c in ctxs + [solo]:
            c.set_option("sor_exact", 1)
        for mode in modes:
            tr = sor_traffic("hs" if name.startswith("hs") else "brox", mode)
            if tr:
                modes[mode]["traffic"] = tr
        modes["exact"]["mode"] = "exact (reference sweep order, bit-identical to the reference)"
        modes["tolerance"]["mode"] = ("sor_exact = 0: " + ("four-colour sweeps, 2 per launch on LDS tiles (k_hs_tile)" if name.startswith("hs") else
                                      "finest level: checkerboard of 64 x 128 tiles, the reference's order inside a tile (k_brox_wave); coarser levels red-black, 4 sweeps per launch on LDS tiles (k_brox_tile)"))
        if ref is not None:
            fn = getattr(ref, ref_name)
            ref.set_num_threads(1)
            t0 = time.perf_counter()
            r1 = fn(I1, I2, **kw)
            t1 = time.perf_counter() - t0
            ref.set_num_threads(8)
            t0 = time.perf_counter()
            r8 = fn(I1, I2, **kw)
            t8 = time.perf_counter() - t0
            ref.set_num_threads(1)

            def cmp(a_, b_):
                return {"aepe": float(np.mean(np.hypot(a_[0] - b_[0], a_[1] - b_[1]))),
                        "max_abs": float(max(np.abs(a_[0] - b_[0]).max(), np.abs(a_[1] - b_[1]).max()))}
            for mode in modes:
                modes[mode]["vs_reference_one_thread"] = dict(cmp(flows[mode], r1), pair="P0", tolerance_aepe=1e-4)
            modes["tolerance"]["reference_8_threads_vs_1"] = dict(cmp(r8, r1), pair="P0",
                                                                  note="the reference's own spread: its in-place sweeps race under OpenMP")
            modes["tolerance"]["reference_seconds"] = {"threads_1": round(t1, 3), "threads_8": round(t8, 3)}
        out[name] = {"size": "%dx%d" % size, "bytes_per_pixel_sweep": bpp}
        out[name].update(modes["exact"])                     # the default mode at the top level, as in round 3
        out[name]["tolerance"] = modes["tolerance"]
        del ins, flo
    for c in ctxs + [solo]:
        c.close()
    return out


def cli_leg(synth, local):
    """What `bin/tvl1flow a.pgm b.pgm out.flo` costs end to end (src/tvl1flow_main.cpp:177-214): a FRESH process per run -- dynamic
    loading, HIP initialisation, context, arena, PGM read, solve, .flo write, teardown -- on the 1080p pair P1 with the headline's
    parameters (strict mode, the front-end's default), images and output on tmpfs.  wall = perf_counter around the child;
    `phases_ms` from the front-end's own clock (OFX_STATS); `outside_main_ms` = wall - the phases = loader + exit."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "optical-flow-1_amd", "bin", "tvl1flow")
    if not os.path.exists(exe):
        return {"error": "bin/tvl1flow not built"}
    nx, ny = 1920, 1080
    I0, I1 = synth.pair("P1", nx, ny)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        for name, img in (("a.pgm", I0), ("b.pgm", I1)):
            with open(os.path.join(d, name), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (nx, ny))
                f.write(np.clip(img, 0, 255).astype(np.uint8).tobytes())
        env = dict(os.environ, OFX_DEVICE=str(local), OFX_STATS=os.path.join(d, "stats.json"))
        args = [exe, os.path.join(d, "a.pgm"), os.path.join(d, "b.pgm"), os.path.join(d, "out.flo"), "0", str(PAR["tau"]), str(PAR["lam"]),
                str(PAR["theta"]), str(PAR["nscales"]), str(PAR["zfactor"]), str(PAR["warps"]), str(PAR["epsilon"]), "0"]
        runs = []
        for _ in range(4):                                   # the first run also pages the binaries in: reported, not used for the median
            t0 = time.perf_counter()
            r = subprocess.run(args, env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode:
                return {"error": "tvl1flow exited %d: %s" % (r.returncode, r.stderr[-300:])}
            st = json.load(open(os.path.join(d, "stats.json")))
            runs.append((wall, st))
        flo_bytes = os.path.getsize(os.path.join(d, "out.flo"))
    warm = sorted(runs[1:], key=lambda x: x[0])
    wall, st = warm[len(warm) // 2]
    ph = st.get("phases_ms", {})
    return {"program": "bin/tvl1flow a.pgm b.pgm out.flo 0 %g %g %g %d %g %d %g 0" % (PAR["tau"], PAR["lam"], PAR["theta"], PAR["nscales"],
                                                                                      PAR["zfactor"], PAR["warps"], PAR["epsilon"]),
            "pair": "P1 1920x1080, 8-bit PGM on tmpfs", "wall_s": round(wall, 4), "wall_s_all_runs": [round(w, 4) for w, _ in runs],
            "phases_ms": {k: round(v, 2) for k, v in ph.items()}, "solve_call_ms": round(st.get("total_ms", 0.0), 2),
            "outside_main_ms": round(wall * 1e3 - sum(ph.values()), 2), "flo_bytes": flo_bytes,
            "note": "fresh process per run; median of the three runs after the first; the solve is the host-array entry point "
                    "(2 x 16.6 MB up, 2 x 16.6 MB down), strict mode"}


def occ_leg(ofx_mod, synth, local):
    """SURVEY 8(f)1 (next row, not the headline): TV-L1 with occlusions, 640x480 triples of the synthetic sequence with the
    reference's defaults (5 levels, 2 warps).  `one_triple`: one solve through the host entry point; `batch`: 48 triples in
    lockstep groups of 16 on 3 contexts (host arrays in / out, so uploads and downloads are inside; result planes reused from
    the warm-up call -- fresh ones would be first touched, i.e. page-faulted, inside the timed call).  work = outer iterations x
    pixels, as the reference's verbose line counts them; every result is bit-identical to the reference on a zero-filled heap
    (tests/test_gpu_occ.py, tests/test_gpu_golden_cli.py)."""
    nx, ny, ns, NB = 640, 480, 5, 48
    kw = dict(lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=ns, zfactor=0.5, warps=2, epsilon=0.01)
    solo = ofx_mod.Ofx(local, ofx_mod.F64)
    ctxs = [ofx_mod.Ofx(local, ofx_mod.F64) for _ in range(3)]
    seq = synth.sequence(nx, ny, 3, 1)
    res = solo.tvl1occ_multiscale(seq[0], seq[1], seq[2], **kw)                          # warm (arena, clocks, result planes)
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        solo.tvl1occ_multiscale(seq[0], seq[1], seq[2], out=res, **kw)
        reps.append(time.perf_counter() - t0)
    dt1 = sorted(reps)[1]                                                                # median of three
    st = solo.stats()
    triples = [tuple(synth.sequence(nx, ny, 3, k + 1)) for k in range(NB)]
    res_b = ofx_mod.tvl1occ_batch(ctxs, triples, **kw)                                   # warm both contexts (arena, result planes)
    t0 = time.perf_counter()
    ofx_mod.tvl1occ_batch(ctxs, triples, out=res_b, **kw)
    dtb = time.perf_counter() - t0
    out = {"size": "%dx%d" % (nx, ny), "levels": ns, "warps": 2,
           "schedule": "ROF box sweeps with all 10 iterations of a call in flight, chi solver 5 iterations per launch (DESIGN 5.6)",
           "one_triple": {"seconds": round(dt1, 4), "repetitions": [round(r, 4) for r in reps], "outer_iterations": int(st.iterations().sum()),
                          "mpix_outer_iters_per_s": round(st.work_pix_iters / dt1 / 1e6, 1)},
           "batch": {"seconds": round(dtb, 4), "triples": NB, "contexts": len(ctxs), "lockstep_group": 16, "ms_per_triple": round(dtb / NB * 1e3, 2)}}
    for c in ctxs + [solo]:
        c.close()
    return out


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    if world > ndev and a.backend == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible (RCCL needs one GPU per rank; "
                         "--backend gloo rehearses several ranks on one GPU)" % (world, ndev))
    local = local % ndev                                 # rehearsal of N ranks on a 1-GPU box (gloo); identity on a real node
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    ofx_mod = importlib.import_module("optical-flow-1_amd")
    synth = importlib.import_module("optical-flow-1_amd.synth")
    batch = importlib.import_module("optical-flow-1_amd.batch")
    prec = ofx_mod.F64 if a.precision == "f64" else ofx_mod.F32
    tdt = torch.float64 if a.precision == "f64" else torch.float32
    nx, ny = WORKLOADS[a.workload]
    nx, ny = a.nx or nx, a.ny or ny
    strong = a.workload == "4k-batch"
    total_steps = a.steps if a.steps > 0 else 64
    if strong:
        mine = batch.pairs_of_rank(total_steps, world, rank)          # pair k of the batch -> rank k mod N
        nsteps = len(mine)
    else:
        nsteps = total_steps
        mine = list(range(rank * nsteps, (rank + 1) * nsteps))
    slots = batch.pairs_per_rank(total_steps, world) if strong else nsteps

    nstreams = max(1, min(a.streams, max(nsteps, 1)))
    ctxs = [ofx_mod.Ofx(local, prec) for _ in range(nstreams)]
    ctx = ctxs[0]
    for c_ in ctxs:
        c_.set_option("concurrency", a.concurrency or nstreams)
        c_.set_option("lockstep", a.lockstep if a.lockstep > 0 else 0)
        if a.rows:
            c_.set_option("rows_per_wave", a.rows)
        if a.chunk:
            c_.set_option("chunk", a.chunk)
        if a.rows2:
            c_.set_option("rows_per_wave2", a.rows2)
        c_.set_option("relaxed_dual", 1 if a.mode == "tolerance" else 0)
        for kv in a.opt:
            c_.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    nrounds = a.rounds if a.rounds > 0 else (2 if (world > 1 and nsteps >= 32) else 1)
    rounds = split_rounds(slots, nrounds, nstreams)          # by slot count: identical on every rank (gather sizes must agree)
    # group size: --lockstep, or what the library picks for the largest round on these contexts; pinned for the whole
    # run so that warmup, timed region and fixed-work pass all use the same grouping
    lockstep = ofx_mod.tvl1_batch_group_size(ctxs, max(max(c for _, c in rounds), 1), nx, ny, PAR["nscales"], PAR["zfactor"])
    for c_ in ctxs:
        c_.set_option("lockstep", lockstep)
    in_flight = nstreams * lockstep

    # ---- inputs, resident in HBM --------------------------------------------------------------------------------
    if strong:
        # every pair of the batch is distinct (P1 with phase 0.3 + 0.1 k, foreground motion (4 + k mod 3, -3)),
        # generated on the device
        nvar = len(mine)
        dI0s, dI1s = [], []
        for k in mine:
            i0, i1 = synth.pair_device(a.pair, nx, ny, k, dev, tdt)
            dI0s.append(i0)
            dI1s.append(i1)
        var_of = lambda i: i
    else:
        # a short synthetic sequence: `variants` distinct pairs cycled, so the pairs of a lockstep group converge differently
        nvar = max(1, a.variants)
        host_pairs = [synth.pair(a.pair, nx, ny, k) for k in range(nvar)]
        dI0s = [torch.from_numpy(p[0]).to(dev, tdt).contiguous() for p in host_pairs]
        dI1s = [torch.from_numpy(p[1]).to(dev, tdt).contiguous() for p in host_pairs]
        var_of = lambda i: (rank * nsteps + i) % nvar
    flo = torch.empty((max(slots, in_flight, 1), ny, nx, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def run_pairs(first, cnt):
        """pairs [first, first + cnt) of this rank through the library's batch entry point: cut into lockstep groups of
        `lockstep` pairs that share every kernel launch; group q runs on context q % nstreams (one host thread + one
        HIP stream each, inside libofx).  Blocking: returns when every flow is in `flo`."""
        if cnt <= 0:
            return 0.0
        idx = range(first, first + cnt)
        work = ofx_mod.tvl1_batch_dev(ctxs, [dI0s[var_of(i)].data_ptr() for i in idx],
                                      [dI1s[var_of(i)].data_ptr() for i in idx],
                                      [flo[i % flo.shape[0]].data_ptr() for i in idx], nx, ny, **PAR)
        return sum(work)

    def fence():
        for c_ in ctxs:
            c_.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log("rank %d/%d: %d pairs %dx%d resident, rounds %s, lockstep %d x %d contexts; warmup"
        % (rank, world, nsteps, nx, ny, [c for _, c in rounds], lockstep, nstreams))
    for i in range(a.warmup):
        run_pairs(0, min(in_flight, max(nsteps, 1)))
    log("warmup done")
    gathered = None
    if world > 1 and rank == 0:
        gathered = [[torch.empty((cnt, ny, nx, 2), dtype=torch.float32, device=dev) for _ in range(world)] for _, cnt in rounds]
    if world > 1:
        # one throw-away gather: RCCL sets up its channels on first use, which is not part of the steady state
        w_ = dist.gather(flo[:1], [torch.empty_like(flo[:1]) for _ in range(world)] if rank == 0 else None, dst=0)
    mark = None
    if os.environ.get("OFX_BENCH_MARK"):
        # tools/trace_budget.py finds the timed repetitions in a rocprofv3 kernel trace between two fills of this 7777-element
        # tensor (profiling runs only; the driver's command does not set the variable)
        mark = torch.empty(7777, dtype=torch.float32, device=dev)
        mark.fill_(1.0)
        torch.cuda.synchronize()

    def timed_region():
        """EXACTLY the K steps of the command, bracketed by barrier + device sync on both sides."""
        fence()
        t0 = time.perf_counter()
        work, pending = 0.0, []
        for r, (first, cnt) in enumerate(rounds):
            work += run_pairs(first, min(cnt, nsteps - first))   # returns with the round's flows complete in HBM
            if world > 1:
                # strong scaling: a rank may own one pair fewer than `slots`; the gather buffers have the same size on
                # every rank (rounds are cut from the slot count), rank 0 ignores the unused tail slots
                pending.append(dist.gather(flo[first:first + cnt], gathered[r] if rank == 0 else None, dst=0, async_op=True))
        t_compute = time.perf_counter() - t0
        for w_ in pending:
            w_.wait()
        fence()
        el = time.perf_counter() - t0
        return el, (el - t_compute) * 1e3, work

    # The region is short (20 pairs = 57 ms), so it is repeated and the MEDIAN repetition is reported (`value`,
    # `ms_per_step`, `gather_ms` all from that one repetition); the spread goes into `repetitions`.
    reps = [timed_region() for _ in range(max(1, a.reps))]
    if world > 1:
        t = torch.tensor([r_[0] for r_ in reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                # a repetition takes as long as its slowest rank
        rep_s = [float(x) for x in t.tolist()]
    else:
        rep_s = [r_[0] for r_ in reps]
    order = sorted(range(len(reps)), key=lambda i: rep_s[i])
    mid = order[(len(order) - 1) // 2]                          # lower median: an actually measured repetition
    elapsed, gather_ms, work = rep_s[mid], reps[mid][1], reps[mid][2]
    if mark is not None:
        mark.fill_(2.0)
        torch.cuda.synchronize()
    log("timed region: %d pairs, %d repetitions %s s -> median %.4f s (exposed gather %.2f ms)"
        % (nsteps, len(reps), [round(x, 4) for x in rep_s], elapsed, gather_ms))

    gather_check = None
    if a.check and world > 1 and rank == 0:
        # every flow is a deterministic function of its pair (lockstep groups are bit-identical to solo solves), so
        # rank 0 can recompute what each rank must have sent
        tmp = torch.empty((ny, nx, 2), dtype=torch.float32, device=dev)
        n_checked = 0
        for q in range(world):
            q_pairs = batch.pairs_of_rank(total_steps, world, q) if strong else list(range(q * nsteps, (q + 1) * nsteps))
            for i, k in enumerate(q_pairs):
                if strong:
                    i0, i1 = synth.pair_device(a.pair, nx, ny, k, dev, tdt)
                else:
                    i0, i1 = dI0s[k % nvar], dI1s[k % nvar]
                ctx.tvl1_multiscale_dev(i0.data_ptr(), i1.data_ptr(), tmp.data_ptr(), nx, ny, **PAR)
                ctx.synchronize()
                r = max(j for j, (f_, _) in enumerate(rounds) if f_ <= i)
                got = gathered[r][q][i - rounds[r][0]]
                if not torch.equal(got.view(torch.int32), tmp.view(torch.int32)):
                    raise SystemExit("gather check FAILED: rank %d slot %d (pair %d)" % (q, i, k))
                n_checked += 1
        gather_check = "ok: %d payloads from %d ranks byte-identical to rank 0's own solves" % (n_checked, world)
        log("gather check " + gather_check)
    if world > 1:
        t = torch.tensor([gather_ms], dtype=torch.float64, device=dev)
        w = torch.tensor([work], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        gather_ms, work = float(t[0].item()), float(w.item())

    # how the loops of this workload end (one lockstep group, untimed): a loop that stops on the first iteration of a
    # fused pair continues from the stored intermediate state (option store_a) or recomputes that iteration
    odd = None
    if rank == 0 and nsteps > 0:
        gsz = min(lockstep, nsteps)
        st_ = ctx.tvl1_group_dev([dI0s[var_of(i)].data_ptr() for i in range(gsz)], [dI1s[var_of(i)].data_ptr() for i in range(gsz)],
                                 [flo[i].data_ptr() for i in range(gsz)], nx, ny, **PAR)
        ctx.synchronize()
        odd = {"loops": gsz * PAR["nscales"] * PAR["warps"], "odd_stops": sum(s_.odd_stops for s_ in st_),
               "served_from_stored_state": sum(s_.odd_stops_stored for s_ in st_),
               "iterations_per_launch_by_level": [int(st_[0].fused[s_]) for s_ in range(PAR["nscales"])],
               "note": "odd_stops = loops that ended inside a launch unit (2 or 3 fused iterations): the unit's first iterations are "
                       "re-run from its input, or -- two-iteration units only -- taken from the stored intermediate state"}

    # ---- the other f64 mode on the same command (same pairs, same grouping, same repetitions) ---------------------------------
    other = None
    other_name = "strict" if a.mode == "tolerance" else "tolerance"
    if world == 1 and nsteps > 0 and not a.no_other_mode:
        for c_ in ctxs:
            c_.set_option("relaxed_dual", 1 if other_name == "tolerance" else 0)
        run_pairs(0, min(in_flight, max(nsteps, 1)))            # warm
        oreps = [timed_region() for _ in range(max(1, a.reps))]
        osec = sorted(r_[0] for r_ in oreps)
        omid = osec[(len(osec) - 1) // 2]
        owork = oreps[0][2]
        other = {"value": round(owork / omid / 1e6, 1), "unit": "Mpix*warp-iters/s", "ms_per_step": round(omid / max(nsteps, 1) * 1e3, 3),
                 "seconds": [round(x, 5) for x in osec], "pix_iters_per_step": owork / max(nsteps, 1),
                 "arithmetic": ARITH[other_name]}
        for c_ in ctxs:
            c_.set_option("relaxed_dual", 1 if a.mode == "tolerance" else 0)
        log("%s mode on the same command: %.1f (median of %s s)" % (other_name, other["value"], other["seconds"]))

    # ---- one pair at a time (BASELINE configs[0] / [1] read literally: what bin/tvl1flow runs) ---------------------------
    single = None
    if rank == 0 and nsteps > 0 and not a.no_single:
        ctx.set_option("concurrency", 1)
        npair = min(nvar, 8)
        ctx.tvl1_multiscale_dev(dI0s[0].data_ptr(), dI1s[0].data_ptr(), flo[0].data_ptr(), nx, ny, **PAR)    # warm
        ctx.synchronize()
        w1, ts = 0.0, []
        for k in range(npair):
            tq0 = time.perf_counter()
            ctx.tvl1_multiscale_dev(dI0s[k].data_ptr(), dI1s[k].data_ptr(), flo[0].data_ptr(), nx, ny, **PAR)
            ctx.synchronize()
            ts.append(time.perf_counter() - tq0)
            w1 += ctx.stats().work_pix_iters
        single = {"device_resident": {"value": round(w1 / sum(ts) / 1e6, 1), "unit": "Mpix*warp-iters/s",
                                      "ms_per_pair": round(sum(ts) / npair * 1e3, 3), "pairs": npair,
                                      "note": "ofx_tvl1_multiscale_dev, one pair in flight, epsilon = 0.01, inputs and .flo payload in HBM"}}
        # the host entry point (host double planes in, host double planes out -- the reference's calling convention):
        # PCIe and the host-side staging are inside; never part of `value`
        hp = [(dI0s[k].double().cpu().numpy(), dI1s[k].double().cpu().numpy()) for k in range(min(npair, 4))]
        outp = (np.zeros((ny, nx)), np.zeros((ny, nx)))          # the caller's result planes, reused from pair to pair
        ctx.tvl1_multiscale(hp[0][0], hp[0][1], out=outp, **PAR)                                             # warm
        w2, th = 0.0, []
        for I0_, I1_ in hp:
            tq0 = time.perf_counter()
            ctx.tvl1_multiscale(I0_, I1_, out=outp, **PAR)
            th.append(time.perf_counter() - tq0)
            w2 += ctx.stats().work_pix_iters
        single["host_entry"] = {"value": round(w2 / sum(th) / 1e6, 1), "unit": "Mpix*warp-iters/s",
                                "ms_per_pair": round(sum(th) / len(hp) * 1e3, 3), "pairs": len(hp),
                                "note": "ofx_tvl1_multiscale, host arrays in / out (2 x %.1f MB up, 2 x %.1f MB down over PCIe; pageable planes -- they copy at "
                                        "the pinned rate on this stack, tools/pcie_pinning.py -- the result planes reused between calls)"
                                        % (nx * ny * 8 / 1e6, nx * ny * 8 / 1e6)}
        ctx.set_option("concurrency", a.concurrency or nstreams)
        del hp
        log("single-pair leg: device-resident %.3f ms, host entry %.3f ms per pair"
            % (single["device_resident"]["ms_per_pair"], single["host_entry"]["ms_per_pair"]))

    # ---- fixed-work pass + roofline (rank 0's numbers are reported; every rank runs it to stay in step) ----
    fixed, roof, roof4k = None, None, None
    if a.fixed_steps > 0 and nsteps > 0:
        # (a) throughput of the fixed-work job with the same number of pairs in flight as the headline
        for c_ in ctxs:
            c_.set_option("fixed_work", 1)
        nfw = min(in_flight, nsteps)
        run_pairs(0, nfw)                                # warm
        fence()
        tq0 = time.perf_counter()
        fw_par = sum(run_pairs(0, nfw) for _ in range(a.fixed_steps))
        fence()
        tq = time.perf_counter() - tq0
        for c_ in ctxs:
            c_.set_option("fixed_work", 0)

        # (b) one pair alone with HIP events around the iteration launches: per-kernel times for the roofline
        def one():
            ctx.tvl1_multiscale_dev(dI0s[0].data_ptr(), dI1s[0].data_ptr(), flo[0].data_ptr(), nx, ny, **PAR)
            return ctx.stats().work_pix_iters
        gsz = min(lockstep, nsteps)

        def grp():
            return ctx.tvl1_group_dev([dI0s[var_of(i)].data_ptr() for i in range(gsz)], [dI1s[var_of(i)].data_ptr() for i in range(gsz)],
                                      [flo[i].data_ptr() for i in range(gsz)], nx, ny, **PAR)
        roof, levels, fw, tf = roofline_of(ctx, one, a.precision, nx, ny, a.fixed_steps, (gsz, grp) if gsz > 1 else None, a.mode)
        ctx.set_option("concurrency", a.concurrency or nstreams)
        log("fixed-work pass: %d steps in %.3f s" % (a.fixed_steps, tf))
        fixed = {"value": round(fw_par / tq / 1e6, 1), "unit": "Mpix*warp-iters/s",
                 "ms_per_step": round(tq / (a.fixed_steps * nfw) * 1e3, 3), "steps": a.fixed_steps * nfw,
                 "pairs_in_flight": nfw, "iterations_per_warp": 300,
                 "single_pair": {"value": round(fw / tf / 1e6, 1), "ms_per_step": round(tf / a.fixed_steps * 1e3, 3)},
                 "levels": levels}
        # (c) the same measurement at 3840x2160 (north_star: ">= 60 % of peak HBM at 4K"), one pair, one pass
        if rank == 0 and world == 1 and not a.no_4k and (nx, ny) != (3840, 2160):
            j0, j1 = synth.pair_device(a.pair, 3840, 2160, 1, dev, tdt)
            f4 = torch.empty((2160, 3840, 2), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

            def one4k():
                ctx.tvl1_multiscale_dev(j0.data_ptr(), j1.data_ptr(), f4.data_ptr(), 3840, 2160, **PAR)
                return ctx.stats().work_pix_iters
            g4 = 4                                       # pairs per launch of the 4K leg (the 4k-batch workload uses 4 .. 16)
            jj = [synth.pair_device(a.pair, 3840, 2160, 2 + k, dev, tdt) for k in range(g4 - 1)]
            ff = [torch.empty((2160, 3840, 2), dtype=torch.float32, device=dev) for _ in range(g4 - 1)]

            def grp4k():
                return ctx.tvl1_group_dev([j0.data_ptr()] + [t[0].data_ptr() for t in jj], [j1.data_ptr()] + [t[1].data_ptr() for t in jj],
                                          [f4.data_ptr()] + [t.data_ptr() for t in ff], 3840, 2160, **PAR)
            roof4k, lv4, fw4, tf4 = roofline_of(ctx, one4k, a.precision, 3840, 2160, 1, (g4, grp4k), a.mode)
            del jj, ff
            roof4k["single_pair_fixed_work"] = {"value": round(fw4 / tf4 / 1e6, 1), "unit": "Mpix*warp-iters/s", "levels": lv4}
            ctx.set_option("concurrency", a.concurrency or nstreams)
            del j0, j1, f4
            log("4K fixed-work pass: %.3f s" % tf4)
        elif (nx, ny) == (3840, 2160):
            roof4k = None

    cli = None
    if rank == 0 and world == 1 and not a.no_cli:
        cli = cli_leg(synth, local)
        log("cli leg done")
    sor = None
    if rank == 0 and world == 1 and not a.no_sor:
        sor = sor_leg(ofx_mod, synth, local, dev, with_cpu=not a.no_cpu)
        log("sor leg done")
    occ = None
    if rank == 0 and world == 1 and not a.no_occ:
        occ = occ_leg(ofx_mod, synth, local)
        log("occ leg done")
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cnx, cny = WORKLOADS["1080p"] if strong else (nx, ny)

        def gpu_flows(k, I0, I1):
            res = {}
            for m_ in ("strict", "tolerance"):
                ctx.set_option("relaxed_dual", 1 if m_ == "tolerance" else 0)
                res[m_] = ctx.tvl1_multiscale(I0, I1, **PAR)
            ctx.set_option("relaxed_dual", 1 if a.mode == "tolerance" else 0)
            return res
        cpu = cpu_baseline(synth, cnx, cny, a.pair, gpu_flows if a.precision == "f64" else None)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    npairs_job = total_steps if strong else nsteps * world
    if strong:
        wl = ("tvl1flow batch of %d distinct synthetic %dx%d pairs (%s, batch variants k = 0..%d), pair k on rank k mod %d, "
              "nscales=5 warps=5 tau=0.25 lambda=0.15 theta=0.3 zfactor=0.5 epsilon=0.01" % (total_steps, nx, ny, a.pair, total_steps - 1, world))
    else:
        wl = ("tvl1flow %dx%d pair (synthetic %s), nscales=5 warps=5 tau=0.25 lambda=0.15 theta=0.3 zfactor=0.5 epsilon=0.01; "
              "one pair per step per GPU" % (nx, ny, a.pair))
    line = {
        "metric": "Mpix*warp-iters/s, TV-L1 %dx%d 5-scale" % (nx, ny),
        "value": round(work / elapsed / 1e6, 1),
        "unit": "Mpix*warp-iters/s",
        "n_gpus": world, "steps": total_steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / max(nsteps if not strong else total_steps, 1) * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": wl, "workload_name": a.workload,
                   "arithmetic_mode": a.mode if a.precision == "f64" else "f32 fast mode", "arithmetic": ARITH[a.mode] if a.precision == "f64" else "float storage",
                   "pairs_per_gpu": slots, "pairs_in_flight_per_gpu": min(in_flight, max(nsteps, 1)), "lockstep_group": lockstep,
                   "streams_per_gpu": nstreams, "distinct_pairs": total_steps if strong else nvar,
                   "rounds_per_gpu": [c for _, c in rounds],
                   "parallelism": "%d GPU(s) x %d HIP stream(s) x lockstep groups of %d pairs; RCCL gather of the .flo payloads "
                                  "per round, overlapped with the next round" % (world, nstreams, lockstep),
                   "pix_iters_per_step": work / max(npairs_job, 1)},
        "pairs_per_s": round(npairs_job / elapsed, 3),
        "gather_ms": round(gather_ms, 3) if world > 1 else 0.0,
    }
    vals = [work / x / 1e6 for x in rep_s]
    line["repetitions"] = {"n": len(rep_s), "seconds": [round(x, 5) for x in rep_s], "reported": "median repetition",
                           "value_min": round(min(vals), 1), "value_max": round(max(vals), 1),
                           "spread_pct": round(100.0 * (max(vals) - min(vals)) / max(min(vals), 1e-9), 2)}
    if other:
        line[other_name] = other
    if single:
        line["single_pair"] = single
    if odd:
        line["loop_ends"] = odd
    if gather_check:
        line["gather_check"] = gather_check
    if fixed:
        line["fixed_work"] = fixed
    if roof:
        line["roofline"] = roof
    if roof4k:
        line["roofline_4k"] = roof4k
    if sor:
        line["sor"] = sor
    if cli:
        line["cli"] = cli
    if occ:
        line["occ"] = occ
    if cpu:
        line["cpu_baseline"] = cpu
    print(json.dumps(line))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
