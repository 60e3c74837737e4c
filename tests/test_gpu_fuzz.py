"""Randomised sizes and parameters (seeded): the C-ABI TV-L1 path against the oracle, and the operators on
ragged sizes.  Everything stays small enough for the oracle to finish in a fraction of a second."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# a longer soak with other draws: OFX_FUZZ_SEED=7 OFX_FUZZ_N=300 OFX_FUZZ_SOR=40 python -m pytest tests/test_gpu_fuzz.py -m gpu
FUZZ_SEED = int(os.environ.get("OFX_FUZZ_SEED", "2026"))
FUZZ_N = int(os.environ.get("OFX_FUZZ_N", "24"))
FUZZ_SOR = int(os.environ.get("OFX_FUZZ_SOR", "4"))
# library options for the whole soak, e.g. OFX_FUZZ_OPTS="fuse3=1" (three iterations per launch at every size), "rof_pipe=0,chi_fuse=0"
FUZZ_OPTS = [o.split("=") for o in os.environ.get("OFX_FUZZ_OPTS", "").split(",") if "=" in o]
FUZZ_DEFAULTS = {"fuse3": 2, "rof_pipe": 1, "chi_fuse": 1, "gauss_fused": 1, "tile": 0, "sor_lds": 1}


@pytest.fixture(autouse=True)
def soak_options(gpu64):
    for name, value in FUZZ_OPTS:
        gpu64.set_option(name, float(value))
    yield
    for name, _ in FUZZ_OPTS:
        gpu64.set_option(name, FUZZ_DEFAULTS.get(name, 0))


def cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nx, ny = int(rng.integers(24, 200)), int(rng.integers(24, 160))
        zf = float(rng.choice([0.5, 0.5, 0.6, 0.75, 0.8]))
        # deepest pyramid whose coarsest level still takes the zoom gaussian (radius (int)(5 sigma)+1 < size)
        sigma = 0.6 * np.sqrt(1.0 / (zf * zf) - 1.0)
        rad = int(5 * sigma) + 1
        ns, w, h = 1, nx, ny
        while ns < 4 and min(w, h) > max(rad, 5) + 1 and min(int(w * zf + 0.5), int(h * zf + 0.5)) > 6:
            w, h = int(w * zf + 0.5), int(h * zf + 0.5)
            ns += 1
        out.append(dict(nx=nx, ny=ny, pair=str(rng.choice(["P0", "P1"])), k=int(rng.integers(0, 5)),
                        kw=dict(tau=float(rng.choice([0.25, 0.2, 0.1])), lam=float(rng.choice([0.15, 0.05, 0.4])),
                                theta=float(rng.choice([0.3, 0.1, 0.5])), nscales=int(rng.integers(1, ns + 1)), zfactor=zf,
                                warps=int(rng.integers(1, 5)), epsilon=float(rng.choice([0.01, 0.05, 0.002, 0.0])))))
    return out


@pytest.mark.parametrize("c", cases(FUZZ_SEED, FUZZ_N), ids=lambda c: "%dx%d-%s-ns%d-w%d-z%g-e%g" % (
    c["nx"], c["ny"], c["pair"], c["kw"]["nscales"], c["kw"]["warps"], c["kw"]["zfactor"], c["kw"]["epsilon"]))
def test_tvl1_random_configurations(gpu64, orc, synth, c):
    I0, I1 = synth.pair(c["pair"], c["nx"], c["ny"], c["k"])
    uo, vo, it_o, err_o = orc.tvl1_multiscale(I0, I1, **c["kw"])
    ug, vg = gpu64.tvl1_multiscale(I0, I1, **c["kw"])
    st = gpu64.stats()
    assert np.array_equal(st.iterations(), it_o)
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9
    assert np.allclose(st.errors(), err_o, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("seed", range(6))
def test_operators_ragged_sizes(gpu64, orc, seed):
    rng = np.random.default_rng(100 + seed)
    nx, ny = int(rng.integers(2, 140)), int(rng.integers(2, 90))
    a, b = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    assert np.array_equal(gpu64.divergence(a, b), orc.divergence(a, b))
    for g, o in zip(gpu64.forward_gradient(a), orc.forward_gradient(a)):
        assert np.array_equal(g, o)
    for g, o in zip(gpu64.centered_gradient(a), orc.centered_gradient(a)):
        assert np.array_equal(g, o)
    for name in ("dxx", "dyy", "dxy"):
        assert np.array_equal(getattr(gpu64, name)(a), getattr(orc, name)(a))
    u, v = rng.standard_normal((ny, nx)) * 5, rng.standard_normal((ny, nx)) * 5
    assert np.array_equal(gpu64.bicubic_warp(a, u, v, True), orc.bicubic_warp(a, u, v, True))
    assert np.array_equal(gpu64.bicubic_warp(a, u, v, False), orc.bicubic_warp(a, u, v, False))
    if min(nx, ny) > 7:
        assert np.array_equal(gpu64.gaussian(a, 0.8), orc.gaussian(a, 0.8))
        assert np.array_equal(gpu64.zoom_out(a, 0.5), orc.zoom_out(a, 0.5))
        assert np.array_equal(gpu64.zoom_in(a, nx + 3, ny + 5), orc.zoom_in(a, nx + 3, ny + 5))


@pytest.mark.parametrize("seed", range(FUZZ_SOR))
def test_sor_random_configurations(gpu64, orc, synth, seed):
    rng = np.random.default_rng((300 if FUZZ_SEED == 2026 else 1000 * FUZZ_SEED) + seed)
    nx, ny = int(rng.integers(20, 90)), int(rng.integers(16, 70))
    I1, I2 = synth.pair("P1", nx, ny, seed)
    z = np.zeros((ny, nx))
    kw = dict(alpha=float(rng.choice([7.0, 20.0, 40.0])), warps=int(rng.integers(1, 4)), TOL=float(rng.choice([1e-4, 1e-3])),
              maxiter=int(rng.choice([5, 150])))
    uo, vo, it_o = orc.hs_single_scale(I1, I2, z, z, **kw)
    ug, vg = gpu64.hs_single_scale(I1, I2, z, z, **kw)
    assert list(gpu64.stats().iterations()[0]) == it_o
    assert np.abs(ug - uo).max() < 1e-11 and np.abs(vg - vo).max() < 1e-11
    kw = dict(alpha=float(rng.choice([50.0, 18.0])), gamma=float(rng.choice([10.0, 0.0, 7.0])), nscales=1, nu=0.5,
              TOL=1e-4, inner=int(rng.integers(1, 3)), outer=int(rng.integers(1, 4)))
    uo, vo, it_o = orc.brox_spatial(I1, I2, **kw)
    ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_o)
    assert np.abs(ug - uo).max() < 1e-10 and np.abs(vg - vo).max() < 1e-10


FUZZ_GROUPS = int(os.environ.get("OFX_FUZZ_GROUPS", "3"))


@pytest.mark.parametrize("seed", range(FUZZ_GROUPS))
def test_tvl1_random_lockstep_groups(gpu64, orc, synth, seed):
    """a lockstep group of random size with random parameters: every pair equals the oracle's solve of that pair"""
    import torch
    rng = np.random.default_rng((500 if FUZZ_SEED == 2026 else 2000 * FUZZ_SEED) + seed)
    c = cases(int(rng.integers(0, 1 << 30)), 1)[0]
    G = int(rng.integers(2, 17))
    nx, ny, kw = c["nx"], c["ny"], dict(c["kw"])
    kw["warps"] = min(kw["warps"], 3)
    pairs = [synth.pair("P0" if k % 4 == 3 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                              [flo[k].data_ptr() for k in range(G)], nx, ny, **kw)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    for k in range(G):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), it), (k, G, nx, ny, kw)
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), (k, G, nx, ny, kw)


FUZZ_TEMPORAL = int(os.environ.get("OFX_FUZZ_TEMPORAL", "2"))


@pytest.mark.parametrize("seed", range(FUZZ_TEMPORAL))
def test_brox_temporal_random_configurations(gpu64, orc, synth, seed):
    """temporal Brox on random small sequences: sweep counts and flows equal the oracle's (reference sweep order)"""
    rng = np.random.default_rng((700 if FUZZ_SEED == 2026 else 3000 * FUZZ_SEED) + seed)
    nx, ny, frames = int(rng.integers(16, 80)), int(rng.integers(16, 60)), int(rng.integers(3, 7))
    I = synth.sequence(nx, ny, frames, int(rng.integers(0, 4)))
    kw = dict(alpha=float(rng.choice([18.0, 30.0])), gamma=float(rng.choice([7.0, 0.0, 3.0])), nscales=int(rng.integers(1, 3)),
              nu=float(rng.choice([0.75, 0.5])), TOL=float(rng.choice([1e-4, 1e-3])), inner=int(rng.integers(1, 3)),
              outer=int(rng.integers(1, 4)))
    ur, vr, it_r = orc.brox_temporal(I, **kw)
    ug, vg = gpu64.brox_temporal(I, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r), (nx, ny, frames, kw)
    assert np.abs(ug - ur).max() < 1e-10 and np.abs(vg - vr).max() < 1e-10, (nx, ny, frames, kw)


FUZZ_SOR_GROUPS = int(os.environ.get("OFX_FUZZ_SOR_GROUPS", "3"))


@pytest.mark.parametrize("seed", range(FUZZ_SOR_GROUPS))
def test_sor_random_lockstep_groups(gpu64, orc, synth, seed):
    """Horn-Schunck and Brox lockstep groups of random size / geometry on hyperplane-major arrays: every pair's sweep table
    equals the oracle's (reference sweep order) and its .flo payload the oracle's flow cast to float32"""
    import torch
    rng = np.random.default_rng((900 if FUZZ_SEED == 2026 else 4000 * FUZZ_SEED) + seed)
    nx, ny, G = int(rng.integers(20, 120)), int(rng.integers(16, 90)), int(rng.integers(2, 17))
    ns = 2 if min(nx, ny) >= 40 else 1
    pairs = [synth.pair("P0" if k % 4 == 3 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ptr = lambda ts: [t.data_ptr() for t in ts]
    for name, val in (("sor_window", int(rng.choice([0, 3, 8]))), ("sor_rows", int(rng.choice([0, 9, 61]))),
                      ("sor_batch", int(rng.choice([0, 7, 40]))), ("sor_spw", int(rng.choice([0, 1, 2, 4])))):
        gpu64.set_option(name, val)
    try:
        hk = dict(alpha=float(rng.choice([7.0, 20.0])), nscales=ns, zfactor=0.5, warps=int(rng.integers(1, 4)),
                  TOL=float(rng.choice([1e-4, 1e-3])), maxiter=int(rng.choice([6, 150])))
        st = gpu64.hs_group_dev(ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(G)], nx, ny, **hk)
        gpu64.synchronize()
        got = flo.cpu().numpy().copy()
        for k in range(G):
            uo, vo, it = orc.hs_pyramidal(pairs[k][0], pairs[k][1], **hk)
            assert np.array_equal(st[k].iterations(), it), ("hs", k, G, nx, ny, hk)
            assert np.abs(got[k][..., 0] - uo).max() < 1e-6 and np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), ("hs", k, G, nx, ny)
        bk = dict(alpha=float(rng.choice([50.0, 18.0])), gamma=float(rng.choice([10.0, 0.0])), nscales=ns, nu=0.5, TOL=1e-4,
                  inner=int(rng.integers(1, 3)), outer=int(rng.integers(1, 4)))
        st = gpu64.brox_group_dev(ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(G)], nx, ny, **bk)
        gpu64.synchronize()
        got = flo.cpu().numpy().copy()
        for k in range(G):
            uo, vo, it = orc.brox_spatial(pairs[k][0], pairs[k][1], **bk)
            assert np.array_equal(st[k].iterations(), it), ("brox", k, G, nx, ny, bk)
            assert np.abs(got[k][..., 0] - uo).max() < 1e-6, ("brox", k, G, nx, ny)
            # the flows agree to < 1e-11; their float32 casts can only differ where a value sits on a rounding boundary
            assert np.mean(got[k] != np.stack([uo, vo], axis=-1).astype(np.float32)) < 1e-3, ("brox", k, G, nx, ny)
    finally:
        for name in ("sor_window", "sor_rows", "sor_batch", "sor_spw"):
            gpu64.set_option(name, 0)


FUZZ_OCC = int(os.environ.get("OFX_FUZZ_OCC", "4"))


@pytest.mark.parametrize("seed", range(FUZZ_OCC))
def test_tvl1occ_random_lockstep_groups(ofx_mod, gpu64, orc, synth, seed):
    """TV-L1 with occlusions: random sizes (rows above / below the 125-row blocks of the ROF sweep, widths below its 24-step
    launch window), parameters and group sizes; every triple of a lockstep group equals the oracle bit for bit"""
    rng = np.random.default_rng(9100 + seed)
    nx, ny = int(rng.integers(20, 150)), int(rng.integers(20, 290))
    ns = 1
    while ns < 3 and min(nx, ny) / 2 ** ns >= 12:
        ns += 1
    kw = dict(lam=float(rng.choice([0.15, 0.3])), alpha=float(rng.choice([0.01, 0.05])), beta=float(rng.choice([0.15, 0.05])),
              theta=float(rng.choice([0.3, 0.2])), nscales=ns, zfactor=0.5, warps=int(rng.integers(1, 3)),
              epsilon=float(rng.choice([0.01, 0.001])))
    G = int(rng.integers(2, 5))
    triples = []
    for k in range(G):
        seq = synth.sequence(nx, ny, 3, int(rng.integers(1, 9)))
        triples.append((seq[0], seq[1], seq[2]) if rng.random() < 0.7 else (seq[1], seq[1], seq[2], seq[0]))
    ctx = ofx_mod.Ofx(0, ofx_mod.F64)
    got = ofx_mod.tvl1occ_batch([ctx], triples, **kw)
    for t, (u, v, c) in zip(triples, got):
        uo, vo, co, _ = orc.tvl1occ_multiscale(t[0], t[1], t[2], filtI0=t[3] if len(t) > 3 else None, **kw)
        assert np.array_equal(u, uo) and np.array_equal(v, vo) and np.array_equal(c, co), (nx, ny, kw)


# ---- round 3: the f64 tolerance mode and robust_expo_methods on random configurations -------------------------------------
@pytest.mark.parametrize("c", cases(FUZZ_SEED + 5, max(6, FUZZ_N // 3)), ids=lambda c: "%dx%d-%s-ns%d-w%d-z%g-e%g" % (
    c["nx"], c["ny"], c["pair"], c["kw"]["nscales"], c["kw"]["warps"], c["kw"]["zfactor"], c["kw"]["epsilon"]))
def test_tolerance_mode_random_configurations(gpu64, orc, synth, c):
    """option relaxed_dual = 1 on random sizes / parameters: north_star's bar (AEPE < 1e-4 px) against the oracle, iteration
    counts within a few per loop (they are equal on every BASELINE config; tiny images sit closer to the threshold)"""
    I0, I1 = synth.pair(c["pair"], c["nx"], c["ny"], c["k"])
    uo, vo, it_o, _ = orc.tvl1_multiscale(I0, I1, **c["kw"])
    gpu64.set_option("relaxed_dual", 1)
    try:
        ug, vg = gpu64.tvl1_multiscale(I0, I1, **c["kw"])
        it_g = gpu64.stats().iterations().copy()
    finally:
        gpu64.set_option("relaxed_dual", 0)
    assert float(np.mean(np.hypot(ug - uo, vg - vo))) < 1e-4
    assert np.abs(it_g - np.asarray(it_o)).max() <= 4


FUZZ_REXPO = int(os.environ.get("OFX_FUZZ_REXPO", "6"))


@pytest.mark.parametrize("seed", range(FUZZ_REXPO))
def test_robust_expo_random_configurations(gpu64, orc, synth, seed):
    rng = np.random.default_rng(FUZZ_SEED + 900 + seed)
    nx, ny = int(rng.integers(20, 150)), int(rng.integers(20, 110))
    ns = int(rng.integers(1, 3)) if min(nx, ny) >= 40 else 1
    kw = dict(method=int(rng.integers(1, 4)), alpha=float(rng.choice([7.9, 18.7, 50.0, 33.3])), gamma=float(rng.choice([0.0, 5.0, 10.0])),
              lam=float(rng.choice([0.05, 0.1, 1.0])), nscales=ns, nu=float(rng.choice([0.5, 0.75])), TOL=float(rng.choice([1e-4, 1e-3])),
              inner=int(rng.integers(1, 3)), outer=int(rng.integers(1, 5)))
    I1, I2 = synth.pair(str(rng.choice(["P0", "P1"])), nx, ny, int(rng.integers(0, 4)))
    uo, vo, it = orc.robust_expo(I1, I2, **kw)
    ug, vg = gpu64.robust_expo(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations()[:ns, :it.shape[1]], it), kw
    assert np.abs(ug - uo).max() < 1e-11 and np.abs(vg - vo).max() < 1e-11, kw


FUZZ_SOR_TOL = int(os.environ.get("OFX_FUZZ_SOR_TOL", "6"))


@pytest.mark.parametrize("seed", range(FUZZ_SOR_TOL))
def test_sor_tolerance_mode_random_lockstep_groups(gpu64, orc, synth, seed):
    """The tolerance-mode sweeps (option sor_exact = 0, csrc/ofx_sor_tile.hip) on random sizes, pyramid depths, groups, sweeps per
    launch, tile geometries, tile widths and tile levels: every pair's sweep table and .flo payload equal the oracle's restatement
    of the same sweep order, bit for bit (the result may depend on NONE of the performance knobs)."""
    import torch
    rng = np.random.default_rng((1700 if FUZZ_SEED == 2026 else 6000 * FUZZ_SEED) + seed)
    nx, ny, G = int(rng.integers(20, 300)), int(rng.integers(16, 200)), int(rng.choice([1, 2, 3, 5, 16]))
    ns = 3 if min(nx, ny) >= 80 else (2 if min(nx, ny) >= 40 else 1)
    pairs = [synth.pair("P0" if k % 4 == 3 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ptr = lambda ts: [t.data_ptr() for t in ts]
    tw, wl = int(rng.choice([16, 50, 64, 128])), int(rng.choice([0, 1, 1, 2]))
    opts = (("sor_exact", 0), ("sor_fuse", int(rng.choice([0, 1, 2, 3, 4]))), ("sor_tile", int(rng.choice([0, 1, 2, 3]))),
            ("sor_tile_w", tw), ("sor_wave_levels", wl), ("sor_wave_p", int(rng.choice([0, 2, 6]))))
    for name, val in opts:
        gpu64.set_option(name, val)
    orc.set_sor_order(1)
    orc.set_sor_tile(tw, 64)
    orc.set_sor_wave_levels(wl)
    try:
        hk = dict(alpha=float(rng.choice([7.0, 20.0])), nscales=ns, zfactor=0.5, warps=int(rng.integers(1, 4)),
                  TOL=float(rng.choice([1e-4, 1e-3])), maxiter=int(rng.choice([5, 6, 7, 150])))
        st = gpu64.hs_group_dev(ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(G)], nx, ny, **hk)
        gpu64.synchronize()
        got = flo.cpu().numpy().copy()
        for k in range(G):
            uo, vo, it = orc.hs_pyramidal(pairs[k][0], pairs[k][1], **hk)
            assert np.array_equal(st[k].iterations(), it), ("hs", k, G, nx, ny, hk, opts)
            assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), ("hs", k, G, nx, ny, opts)
        bk = dict(alpha=float(rng.choice([50.0, 18.0])), gamma=float(rng.choice([10.0, 0.0])), nscales=ns, nu=0.5, TOL=1e-4,
                  inner=int(rng.integers(1, 3)), outer=int(rng.integers(1, 4)))
        st = gpu64.brox_group_dev(ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(G)], nx, ny, **bk)
        gpu64.synchronize()
        got = flo.cpu().numpy().copy()
        for k in range(G):
            uo, vo, it = orc.brox_spatial(pairs[k][0], pairs[k][1], **bk)
            assert np.array_equal(st[k].iterations(), it), ("brox", k, G, nx, ny, bk, opts)
            assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), ("brox", k, G, nx, ny, opts)
    finally:
        orc.set_sor_order(0)
        orc.set_sor_tile(128, 64)
        orc.set_sor_wave_levels(0)
        for name, val in (("sor_exact", 1), ("sor_fuse", 0), ("sor_tile", 0), ("sor_tile_w", 0), ("sor_wave_levels", 1), ("sor_wave_p", 0)):
            gpu64.set_option(name, val)
