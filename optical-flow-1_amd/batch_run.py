#!/usr/bin/env python3
"""Batch front-end: many image pairs through TV-L1, sharded over the GPUs of one node (BASELINE config 5).

    # one GPU
    python optical-flow-1_amd/batch_run.py --list pairs.txt --out-dir flows/
    # N GPUs: one process per GPU, RCCL over xGMI for the single end-of-batch gather
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        optical-flow-1_amd/batch_run.py --synthetic 64 --size 3840x2160 --out-dir flows/

`pairs.txt` holds one pair per line: "I0 I1 [out.flo]" (PGM / PPM / PFM / PNG, read exactly as the tvl1flow front-end
reads them).  Pair k is solved by rank k % world (optical-flow-1_amd/batch.py); every rank keeps `--in-flight`
contexts (HIP stream + host thread each) busy on its GPU, every context solving lockstep groups of up to 16 pairs
(ofx_tvl1_batch_dev); the float32 .flo payloads stay in
HBM until ONE gather to rank 0 at the end, which writes the files.  Solver parameters are the reference's
(tvl1flow_main.cpp:24-33), including the automatic number of scales.
"""
import argparse
import ctypes as C
import importlib
import math
import os
import struct
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", help="text file: I0 I1 [out.flo] per line")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic P1 batch variants instead of files")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--tau", type=float, default=0.25)
    ap.add_argument("--lambda", dest="lam", type=float, default=0.15)
    ap.add_argument("--theta", type=float, default=0.3)
    ap.add_argument("--nscales", type=int, default=100)
    ap.add_argument("--zfactor", type=float, default=0.5)
    ap.add_argument("--nwarps", type=int, default=5)
    ap.add_argument("--epsilon", type=float, default=0.01)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--in-flight", type=int, default=4)
    ap.add_argument("--backend", default="nccl")
    return ap.parse_args()


def read_image(io, path):
    w, h = C.c_int(), C.c_int()
    p = io.ofx_read_image_double(path.encode(), C.byref(w), C.byref(h))
    if not p:
        raise SystemExit('ERROR: could not read image from file "%s"' % path)
    return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()


def write_flo(path, uv):
    h, w, _ = uv.shape
    with open(path, "wb") as f:
        f.write(b"PIEH" + struct.pack("<II", w, h))
        f.write(np.ascontiguousarray(uv, dtype=np.float32).tobytes())


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if not torch.cuda.is_available():
        raise SystemExit("batch_run.py needs a GPU")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    ofx = importlib.import_module("optical-flow-1_amd")
    batch = importlib.import_module("optical-flow-1_amd.batch")
    synth = importlib.import_module("optical-flow-1_amd.synth")

    # ---- the batch -------------------------------------------------------------------------------------
    if a.synthetic:
        nx, ny = map(int, a.size.split("x"))
        n_pairs = a.synthetic
        names = ["pair%03d.flo" % k for k in range(n_pairs)]
        load = lambda k: synth.pair("P1", nx, ny, k)
    else:
        if not a.list:
            raise SystemExit("give --list or --synthetic")
        io = C.CDLL(os.path.join(HERE, "libofxio.so"))
        io.ofx_read_image_double.restype = C.POINTER(C.c_double)
        io.ofx_read_image_double.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rows = [ln.split() for ln in open(a.list) if ln.strip() and not ln.startswith("#")]
        n_pairs = len(rows)
        names = [r[2] if len(r) > 2 else "pair%03d.flo" % k for k, r in enumerate(rows)]
        load = lambda k: (read_image(io, rows[k][0]), read_image(io, rows[k][1]))
        first = load(0)[0] if n_pairs else np.zeros((1, 1))
        ny, nx = first.shape
    mine = batch.pairs_of_rank(n_pairs, world, rank)
    slots = batch.pairs_per_rank(n_pairs, world)

    # the front-end's rule for the number of scales (tvl1flow_main.cpp:185-188)
    nscales = a.nscales if a.nscales > 0 else 100
    N = 1 + math.log(math.hypot(nx, ny) / 16.0) / math.log(1 / a.zfactor)
    if N < nscales:
        nscales = int(N)

    tdt = torch.float64 if a.precision == "f64" else torch.float32
    d0, d1 = [], []
    for k in mine:
        I0, I1 = load(k)
        if I0.shape != (ny, nx) or I1.shape != (ny, nx):
            raise SystemExit("ERROR: pair %d: all images of a batch must be %dx%d" % (k, nx, ny))
        d0.append(torch.from_numpy(I0).to(dev, tdt).contiguous())
        d1.append(torch.from_numpy(I1).to(dev, tdt).contiguous())
    flo = torch.zeros((max(slots, 1), ny, nx, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    nctx = max(1, min(a.in_flight, len(mine)))
    ctxs = [ofx.Ofx(local, ofx.F64 if a.precision == "f64" else ofx.F32) for _ in range(nctx)]
    for c in ctxs:
        c.set_option("concurrency", nctx)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    work = ofx.tvl1_batch_dev(ctxs, [t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                              [flo[i].data_ptr() for i in range(len(mine))], nx, ny, tau=a.tau, lam=a.lam, theta=a.theta,
                              nscales=nscales, zfactor=a.zfactor, warps=a.nwarps, epsilon=a.epsilon) if mine else []
    flows = batch.gather_flows(flo, n_pairs, world, rank, dist if world > 1 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(sum(work))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        if a.out_dir:
            os.makedirs(a.out_dir, exist_ok=True)
            for k in range(n_pairs):
                write_flo(os.path.join(a.out_dir, os.path.basename(names[k])), flows[k].cpu().numpy())
        print("%d pairs %dx%d on %d GPU(s), %d scales: %.3f s (%.1f pairs/s, %.0f Mpix*warp-iters/s)%s"
              % (n_pairs, nx, ny, world, nscales, dt, n_pairs / dt, tot.item() / dt / 1e6,
                 ", .flo files in " + a.out_dir if a.out_dir else ""))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
