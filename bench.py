#!/usr/bin/env python3
"""bench.py -- TV-L1 multiscale throughput on MI355X, BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one synthetic 1920x1080 image pair (P1 of SURVEY.md §8d, rank r uses batch variant k=r)
through ofx_tvl1_multiscale_dev with the reference's default parameters (nscales=5 warps=5 tau=0.25
lambda=0.15 theta=0.3 zfactor=0.5 epsilon=0.01) in f64 storage, inputs already resident in HBM, result
= the .flo payload in HBM.  Work is counted exactly like the reference prints it:
    work = sum_scales sum_warps n_iter * nx_s * ny_s      [pixel-iterations]
and `value` = whole-job Mpix*warp-iters/s = (work of all ranks over the K timed steps) / (max over ranks
of the wall time of those K steps, barrier + device sync on both sides) / 1e6.  For N > 1 every rank
processes its own K pairs (weak scaling, no data-path collective) and the timed region ends with ONE RCCL
gather of the float32 .flo payloads to rank 0.

Because P1 converges in a few dozen iterations per warp, the line also carries a `fixed_work` object
(option "fixed_work": every warp runs exactly 300 iterations) and the `roofline` object, which
is measured on that pass: HIP events on the library's own stream bracket the full-resolution iteration
launches, achieved = 120 B/px/iter * 1920*1080 px / average launch duration.  `cpu_baseline` times the
compiled reference (oracle/_ref, kind "reference"; falls back to the C port) on the host cores, rank 0 and
N=1 only.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY = 1920, 1080
PAR = dict(tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5, epsilon=0.01)
BYTES_PER_PIX_ITER = {0: 120.0, 1: 60.0}      # 15 storage elements / px / iteration
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--nx", type=int, default=NX)
    ap.add_argument("--ny", type=int, default=NY)
    ap.add_argument("--pair", default="P1")
    ap.add_argument("--fixed-steps", type=int, default=2, help="fixed-work passes for the roofline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--lockstep", type=int, default=0,
                    help="pairs per lockstep group (they share every kernel launch of one context); 0 = the library's "
                         "choice: up to 4, fewer when the batch is small")
    ap.add_argument("--concurrency", type=int, default=0, help="override the library's concurrency hint (0 = streams)")
    ap.add_argument("--variants", type=int, default=8, help="distinct synthetic pairs cycled through by the steps")
    ap.add_argument("--streams", type=int, default=4,
                    help="image pairs in flight per GPU, each on its own context / HIP stream (SURVEY 8e: >= 2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--rows2", type=int, default=0)
    return ap.parse_args()


T_START = time.perf_counter()


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def cpu_baseline(synth, nx, ny, pair):
    """Reference (oracle/_ref) on the host cores: one multiscale call on the same pair + the inner loop only."""
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
    # a bounded sample (~10 s of CPU work): NPAIR batch variants of the bench pair through the multiscale call
    NPAIR = 4
    pairs = [synth.pair(pair, nx, ny, k) for k in range(NPAIR)]
    t0 = time.perf_counter()
    for I0, I1 in pairs:
        cpu.tvl1_multiscale(I0, I1, **PAR)
    t_ms = time.perf_counter() - t0
    log("cpu_baseline: %d multiscale calls %.2f s" % (NPAIR, t_ms))
    out = {"unit": "Mpix*warp-iters/s", "cores": cores, "kind": cpu.kind,
           "sample": "%d pairs %s %dx%d (batch variants 0..%d), same parameters, Dual_TVL1_optic_flow_multiscale (%.2f s)"
                     % (NPAIR, pair, nx, ny, NPAIR - 1, t_ms), "seconds": round(t_ms, 3),
           "build": "-O3 -fopenmp, generic x86-64"}
    # work of those calls: iteration counts from the port (bit-identical loop, OMP_NUM_THREADS=1 parity-tested)
    work = None
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
    # fixed-work inner loop: Dual_TVL1_optic_flow, 1 warp, eps=0 -> 300 iterations at full resolution
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
    return out, work


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()      # rehearsal of N ranks on a 1-GPU box (gloo); identity on a real node
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
    prec = ofx_mod.F64 if a.precision == "f64" else ofx_mod.F32
    tdt = torch.float64 if a.precision == "f64" else torch.float32
    nstreams = max(1, min(a.streams, max(a.steps, 1)))
    ctxs = [ofx_mod.Ofx(local, prec) for _ in range(nstreams)]
    ctx = ctxs[0]
    nx, ny = a.nx, a.ny
    for c_ in ctxs:
        c_.set_option("concurrency", a.concurrency or nstreams)
        c_.set_option("lockstep", a.lockstep if a.lockstep > 0 else 0)
        if a.rows:
            c_.set_option("rows_per_wave", a.rows)
        if a.chunk:
            c_.set_option("chunk", a.chunk)
        if a.rows2:
            c_.set_option("rows_per_wave2", a.rows2)
    # group size: --lockstep, or what the library picks for a batch of `steps` pairs on these contexts; pinned
    # for the whole run so that warmup, timed region and fixed-work pass all use the same grouping
    lockstep = ofx_mod.tvl1_batch_group_size(ctxs, max(a.steps, 1), nx, ny, PAR["nscales"], PAR["zfactor"])
    for c_ in ctxs:
        c_.set_option("lockstep", lockstep)
    in_flight = nstreams * lockstep

    # a short synthetic sequence: `variants` distinct pairs (SURVEY 8d batch variants), resident in HBM; step i of
    # rank r solves variant (r * steps + i) % variants, so the pairs of a lockstep group converge differently
    nvar = max(1, a.variants)
    host_pairs = [synth.pair(a.pair, nx, ny, k) for k in range(nvar)]
    dI0s = [torch.from_numpy(p[0]).to(dev, tdt).contiguous() for p in host_pairs]
    dI1s = [torch.from_numpy(p[1]).to(dev, tdt).contiguous() for p in host_pairs]
    var_of = lambda i: (rank * max(a.steps, 1) + i) % nvar
    flo = torch.empty((max(a.steps, in_flight, 1), ny, nx, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step(i, c_=None):
        c_ = c_ or ctx
        v = var_of(i)
        c_.tvl1_multiscale_dev(dI0s[v].data_ptr(), dI1s[v].data_ptr(), flo[i % flo.shape[0]].data_ptr(), nx, ny, **PAR)
        return c_.stats().work_pix_iters

    def run_steps(n):
        """n steps (pairs) through the library's batch entry point: the pairs are cut into lockstep groups of
        `lockstep` pairs that share every kernel launch; group q runs on context q % nstreams (one host thread +
        one HIP stream each, inside libofx), so while one group waits for a convergence poll another one fills
        the GPU."""
        if in_flight == 1:
            return sum(step(i) for i in range(n))
        work = ofx_mod.tvl1_batch_dev(ctxs, [dI0s[var_of(i)].data_ptr() for i in range(n)],
                                      [dI1s[var_of(i)].data_ptr() for i in range(n)],
                                      [flo[i % flo.shape[0]].data_ptr() for i in range(n)], nx, ny, **PAR)
        return sum(work)

    def fence():
        for c_ in ctxs:
            c_.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log("rank %d/%d: inputs resident, warmup" % (rank, world))
    for i in range(a.warmup):
        run_steps(in_flight)
    log("warmup done")
    gathered = None
    if world > 1 and rank == 0:
        gathered = [torch.empty_like(flo) for _ in range(world)]
    fence()
    t0 = time.perf_counter()
    work = run_steps(a.steps)
    for c_ in ctxs:
        c_.synchronize()
    if world > 1:
        dist.gather(flo, gathered, dst=0)       # the one collective: .flo payloads to rank 0 over xGMI (RCCL)
    fence()
    elapsed = time.perf_counter() - t0
    log("timed region: %d steps in %.3f s" % (a.steps, elapsed))

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        w = torch.tensor([work], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        elapsed, work = float(t.item()), float(w.item())

    # ---- fixed-work pass + roofline (rank 0's numbers are reported; every rank runs it to stay in step) ----
    fixed, roof = None, None
    if a.fixed_steps > 0:
        # (a) throughput of the fixed-work job with the same number of pairs in flight as the headline
        for c_ in ctxs:
            c_.set_option("fixed_work", 1)
        run_steps(in_flight)                                 # warm
        fence()
        tq0 = time.perf_counter()
        fw_par = run_steps(a.fixed_steps * in_flight)
        fence()
        tq = time.perf_counter() - tq0
        # (b) one pair alone with HIP events around the iteration launches: per-kernel times for the roofline
        ctx.set_option("concurrency", 1)
        ctx.set_option("profile", 1)
        step(0)                                              # warm
        ctx.synchronize()
        tf0 = time.perf_counter()
        fw, it_ms, it_n = 0.0, 0.0, 0
        lv_ms, lv_n = [0.0] * PAR["nscales"], [0] * PAR["nscales"]
        for i in range(a.fixed_steps):
            fw += step(0)
            st = ctx.stats()
            it_ms += st.iter_ms[0]
            it_n += st.iter_launches[0]
            for s_ in range(PAR["nscales"]):
                lv_ms[s_] += st.iter_ms[s_]
                lv_n[s_] += st.iter_launches[s_]
        ctx.synchronize()
        tf = time.perf_counter() - tf0
        for c_ in ctxs:
            c_.set_option("fixed_work", 0)
        ctx.set_option("profile", 0)
        ctx.set_option("concurrency", a.concurrency or nstreams)
        log("fixed-work pass: %d steps in %.3f s" % (a.fixed_steps, tf))
        fixed = {"value": round(fw_par / tq / 1e6, 1), "unit": "Mpix*warp-iters/s",
                 "ms_per_step": round(tq / (a.fixed_steps * in_flight) * 1e3, 3), "steps": a.fixed_steps * in_flight,
                 "pairs_in_flight": in_flight, "iterations_per_warp": 300,
                 "single_pair": {"value": round(fw / tf / 1e6, 1), "ms_per_step": round(tf / a.fixed_steps * 1e3, 3)},
                 "levels": [{"size": "%dx%d" % (st.nx[s_], st.ny[s_]), "iter_us": round(lv_ms[s_] * 1e3 / max(lv_n[s_], 1), 2),
                             "ms_per_step": round(lv_ms[s_] / a.fixed_steps, 2)} for s_ in range(PAR["nscales"])]}
        us = it_ms * 1e3 / max(it_n, 1)                       # per ITERATION (stats count iterations)
        ach = BYTES_PER_PIX_ITER[prec] * nx * ny / (us * 1e-6) / 1e9
        # the dominant kernel is k_tvl1_iter2: TWO iterations per launch
        roof = {"bound": "hbm", "kernel": "k_tvl1_iter2<%s> @ %dx%d (2 fused iterations per launch)"
                                          % ("double" if prec == 0 else "float", nx, ny),
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_launch_us": round(2 * us, 3), "launches": int(it_n // 2),
                "algorithmic_bytes_per_launch": 2 * BYTES_PER_PIX_ITER[prec] * nx * ny,
                "algorithmic_bytes_per_pixel_iteration": BYTES_PER_PIX_ITER[prec],
                "mpix_iters_per_s": round(nx * ny / us, 1)}
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                roof["traffic"] = json.load(open(prof)).get("bytes_per_launch_%s_%dx%d" % (a.precision, nx, ny))
            except Exception:
                pass

    cpu, _ = (None, None)
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu, _ = cpu_baseline(synth, nx, ny, a.pair)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    line = {
        "metric": "Mpix*warp-iters/s, TV-L1 %dx%d 5-scale" % (nx, ny),
        "value": round(work / elapsed / 1e6, 1),
        "unit": "Mpix*warp-iters/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": "tvl1flow %dx%d pair (synthetic %s), nscales=5 warps=5 tau=0.25 lambda=0.15 theta=0.3 "
                               "zfactor=0.5 epsilon=0.01; one pair per step per GPU" % (nx, ny, a.pair),
                   "pairs_per_gpu": a.steps, "pairs_in_flight_per_gpu": in_flight, "lockstep_group": lockstep,
                   "streams_per_gpu": nstreams, "distinct_pairs": nvar,
                   "parallelism": "%d GPU(s) x %d HIP stream(s) x lockstep groups of %d pairs, RCCL gather of .flo at end"
                                  % (world, nstreams, lockstep),
                   "pix_iters_per_step": work / max(a.steps, 1) / world},
        "pairs_per_s": round(a.steps * world / elapsed, 3),
    }
    if fixed:
        line["fixed_work"] = fixed
    if roof:
        line["roofline"] = roof
    if cpu:
        line["cpu_baseline"] = cpu
    print(json.dumps(line))


if __name__ == "__main__":
    main()
