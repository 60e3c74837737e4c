"""Error behaviour of the C ABI on a real device: bad arguments are reported as status codes (the
reference would crash or throw), and a failed call leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_argument_validation(gpu64, ofx_mod, synth):
    L = ofx_mod.lib()
    I0, I1 = synth.pair("P0", 64, 48)
    u, v = np.zeros((48, 64)), np.zeros((48, 64))
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    raw = C.CDLL(ofx_mod.LIB_PATH)
    f = raw.ofx_tvl1_multiscale
    f.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int,
                                    C.c_double, C.c_int]
    args = (48 * 0 + 64, 48, 0.25, 0.15, 0.3, 3, 0.5, 5, 0.01, 0)
    assert f(gpu64.h, None, dp(I1), dp(u), dp(v), *args) == 1                       # NULL image
    assert f(None, dp(I0), dp(I1), dp(u), dp(v), *args) == 1                        # NULL context
    assert f(gpu64.h, dp(I0), dp(I1), dp(u), dp(v), 1, 48, 0.25, 0.15, 0.3, 1, 0.5, 5, 0.01, 0) == 1    # nx < 2
    assert f(gpu64.h, dp(I0), dp(I1), dp(u), dp(v), 64, 48, 0.25, 0.15, 0.3, 0, 0.5, 5, 0.01, 0) == 1   # nscales 0
    assert f(gpu64.h, dp(I0), dp(I1), dp(u), dp(v), 64, 48, 0.25, 0.15, 0.3, 3, 1.5, 5, 0.01, 0) == 1   # zfactor
    assert f(gpu64.h, dp(I0), dp(I1), dp(u), dp(v), 64, 48, 0.25, 0.15, 0.3, 3, 0.5, 0, 0.01, 0) == 1   # warps 0
    assert f(gpu64.h, dp(I0), dp(I1), dp(u), dp(v), 64, 48, 0.25, 0.15, 0.3, 8, 0.5, 5, 0.01, 0) == 2   # pyramid too deep
    assert b"sigma" in L.ofx_last_error(gpu64.h).lower() or b"scale" in L.ofx_last_error(gpu64.h).lower()
    with pytest.raises(ofx_mod.OfxError) as e:
        gpu64.set_option("no_such_option", 1)
    assert e.value.status == 1
    with pytest.raises(ofx_mod.OfxError):
        gpu64.gaussian(np.zeros((8, 8)), -1.0)
    with pytest.raises(ofx_mod.OfxError):
        gpu64.hs_single_scale(I0, I1, u, v, maxiter=10 ** 6)
    # the context still works after all those failures
    uu, vv = gpu64.tvl1_multiscale(I0, I1, nscales=3)
    assert np.isfinite(uu).all() and gpu64.stats().nscales == 3


def test_context_reuse_across_sizes_and_solvers(gpu64, orc, synth):
    """the arena is reset per call and coalesced once: interleaving sizes / solvers must not disturb results"""
    ref = {}
    for nx, ny in ((64, 48), (160, 120), (96, 64)):
        I0, I1 = synth.pair("P1", nx, ny)
        ref[(nx, ny)] = orc.tvl1_multiscale(I0, I1, nscales=3)[:2]
    for rep in range(2):
        for (nx, ny), (uo, vo) in ref.items():
            I0, I1 = synth.pair("P1", nx, ny)
            u, v = gpu64.tvl1_multiscale(I0, I1, nscales=3)
            assert np.abs(u - uo).max() < 1e-9 and np.abs(v - vo).max() < 1e-9
            gpu64.hs_pyramidal(I0, I1, alpha=20.0, nscales=2, warps=2)
            gpu64.divergence(I0, I1)


def test_two_contexts_from_two_threads(ofx_mod, orc, synth):
    """calls on different contexts are independent (this is how bench.py keeps 4 pairs in flight)"""
    import threading
    I0, I1 = synth.pair("P1", 200, 150)
    uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=4)
    out = [None, None]

    def work(w):
        c = ofx_mod.Ofx(0, ofx_mod.F64)
        for _ in range(3):
            out[w] = c.tvl1_multiscale(I0, I1, nscales=4)
        c.close()
    th = [threading.Thread(target=work, args=(w,)) for w in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for u, v in out:
        assert np.abs(u - uo).max() < 1e-9 and np.abs(v - vo).max() < 1e-9


def test_batch_entry_point(ofx_mod, orc, synth):
    """ofx_tvl1_batch_dev: 5 different pairs over 2 contexts == each pair solved alone"""
    import torch
    nx, ny, n = 128, 96, 5
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(2)]
    pairs = [synth.pair("P1", nx, ny, k) for k in range(n)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((n, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    work = ofx_mod.tvl1_batch_dev(ctxs, [t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                  [flo[k].data_ptr() for k in range(n)], nx, ny, nscales=3)
    got = flo.cpu().numpy()
    for k in range(n):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], nscales=3)
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32))
        sizes = [(nx, ny), (64, 48), (32, 24)]
        assert work[k] == sum(int(it[s].sum()) * sizes[s][0] * sizes[s][1] for s in range(3))
    # the automatic grouping: as few rounds as groups of 16 allow, evened out over the contexts
    size = lambda npairs, k: ofx_mod.tvl1_batch_group_size(ctxs[:k], npairs, nx, ny, 3, 0.5)
    assert [size(5, 2), size(1, 2), size(20, 2), size(33, 2), size(64, 2), size(65, 2), size(7, 1)] == [3, 1, 10, 9, 16, 11, 7]
    ctxs[0].set_option("lockstep", 2)
    assert size(20, 2) == 2
    ctxs[0].set_option("lockstep", 0)
    for c in ctxs:
        c.close()


def test_group_size_reports_errors_and_honours_the_memory_budget(ofx_mod):
    """ofx_tvl1_batch_group_size returns -status on error (not a positive status that reads as a group size),
    and its memory cap uses the real footprint of a pair (26 elements per pixel and level): option mem_budget"""
    L = ofx_mod.lib()
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(2)]
    arr = (C.c_void_p * 2)(*[c.h.value for c in ctxs])
    assert L.ofx_tvl1_batch_group_size(arr, 2, 8, 0, 48, 3, 0.5) == -1              # nx = 0 -> -OFX_ERR_ARG
    assert L.ofx_tvl1_batch_group_size(arr, 2, 8, 64, 48, 8, 0.5) == -2             # pyramid too deep -> -OFX_ERR_SIGMA
    assert L.ofx_tvl1_batch_group_size(None, 2, 8, 64, 48, 3, 0.5) == -1
    with pytest.raises(ofx_mod.OfxError) as e:
        ofx_mod.tvl1_batch_group_size(ctxs, 8, 64, 48, 8, 0.5)
    assert e.value.status == -2
    # the batch call maps the negative group size back to the status
    import torch
    t = torch.zeros((48, 64), dtype=torch.float64, device="cuda")
    f = torch.zeros((48, 64, 2), dtype=torch.float32, device="cuda")
    with pytest.raises(ofx_mod.OfxError) as e:
        ofx_mod.tvl1_batch_dev(ctxs, [t.data_ptr()] * 4, [t.data_ptr()] * 4, [f.data_ptr()] * 4, 64, 48, nscales=8)
    assert e.value.status == 2
    # 1920x1080, 5 scales: one pair = 26 * 8 B * 2.76 Mpix = 575 MB of level arrays; a budget of 3 GB for 2 contexts
    # leaves (1.5 GB - temporaries) / 575 MB = 2 pairs per group, 1.2 GB just one
    size = lambda: ofx_mod.tvl1_batch_group_size(ctxs, 64, 1920, 1080, 5, 0.5)
    assert size() == 16
    for budget, want in ((3.0e9, 2), (1.2e9, 1), (40e9, 16)):
        ctxs[0].set_option("mem_budget", budget)
        assert size() == want, budget
    ctxs[0].set_option("mem_budget", 0)
    for c in ctxs:
        c.close()


def test_failed_context_creation_cleans_up(ofx_mod):
    """a failing ofx_ctx_create must not leak; creating and destroying many contexts must not either"""
    import torch
    L = ofx_mod.lib()
    h = C.c_void_p()
    assert L.ofx_ctx_create(C.byref(h), 99, 0) == 5 and not h.value                 # no such device
    assert L.ofx_ctx_create(C.byref(h), 0, 7) == 1 and not h.value                  # bad precision
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        c = ofx_mod.Ofx(0, ofx_mod.F64)
        c.tvl1_multiscale(np.zeros((48, 64)), np.zeros((48, 64)), nscales=2)
        c.close()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20
