"""GPU parity of the TV-L1 path against the oracle (which is bit-checked against the compiled
reference in test_oracle_vs_ref.py).

Tolerance (BASELINE.json north_star): average end-point-error delta < 1e-4 px.  In f64 storage the
kernels reproduce the reference's per-pixel IEEE arithmetic exactly; the only non-identical quantity
is the order of the convergence-error sum, so iteration counts must match and the flow is expected
to be (and asserted) equal to ~1e-9."""
import numpy as np
import pytest
from conftest import aepe, require_or_skip

pytestmark = pytest.mark.gpu

PAR = dict(tau=0.25, lam=0.15, theta=0.3)

# Option "tile" = 4 | 6 runs levels of at most 200 000 pixels x pairs with K iterations per launch on 2-D tiles (k_tvl1_tile)
# instead of the marching-strip kernels (default, tile = 0: the tile kernel measured no faster).  Most images of this file are
# small enough for the tile kernel at every level, so every test runs twice -- tile = 0 and tile = 4 -- except the full-size
# ones, which run with the default only.
FULL_SIZE_ONLY_DEFAULT = ("test_full_size", "test_4k_size", "test_headline_launch_shape", "test_roofline_4k_launch_shape",
                          "test_batch_dev_on_four_contexts", "test_f32_mode_is_as_accurate", "test_tile_kernel")


@pytest.fixture(autouse=True, params=[0, 4, -3, -4], ids=["strips", "tile4", "fuse3", "fuse3_fixed_units"])
def tile_mode(request, gpu64, gpu32):
    """which iteration kernel: the marching strips with two iterations per launch (default), the 2-D tile kernel (tile = 4), the
    marching strips with three iterations per launch (fuse3 = 1) as a cursor loop (units shrink near the end of a loop, the
    default) and with fixed units of three (fuse3_cursor = 0)"""
    if request.node.name.startswith(FULL_SIZE_ONLY_DEFAULT):
        if request.param != 0:
            pytest.skip("full-size case: default kernel choice only")
        yield 0                                              # the library's own choice (fuse3 = 2: by mode and group size)
        return
    for c in (gpu64, gpu32):
        c.set_option("tile", request.param if request.param > 0 else 0)
        c.set_option("fuse3", 1 if request.param in (-3, -4) else 0)
        c.set_option("fuse3_cursor", 0 if request.param == -4 else 1)
    yield request.param
    for c in (gpu64, gpu32):
        c.set_option("tile", 0)
        c.set_option("fuse3", 2)
        c.set_option("fuse3_cursor", 1)


@pytest.mark.parametrize("afac1,afac2", [(1.2, 1.5), (1e-9, 1e-9), (1e9, 1e9), (1e-9, 1e9), (1.05, 3.0)])
def test_cursor_loop_unit_sizes_do_not_change_the_result(gpu64, orc, synth, afac1, afac2, tile_mode):
    """k_tvl1_iter3 as a cursor loop: when a launch runs 3, 2 or 1 iterations is a performance decision (error / threshold ratios
    afac2, afac1) -- never 2 or 1 (1e-9), always 1 (1e9), always 2, odd ratios: iteration tables and flows equal the oracle's in all
    of them, for a lone pair and a lockstep group whose pairs stop at different iterations."""
    if tile_mode != -3:
        pytest.skip("cursor loop only")
    import torch
    gpu64.set_option("fuse3_afac1", afac1)
    gpu64.set_option("fuse3_afac2", afac2)
    try:
        nx, ny, G = 150, 97, 5
        pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
        want = [orc.tvl1_multiscale(p[0], p[1], nscales=3, **PAR) for p in pairs]
        u, v = gpu64.tvl1_multiscale(pairs[0][0], pairs[0][1], nscales=3, **PAR)
        assert np.array_equal(gpu64.stats().iterations(), want[0][2])
        assert np.abs(u - want[0][0]).max() < 1e-9 and np.abs(v - want[0][1]).max() < 1e-9
        d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
        d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
        flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
        st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)], nx, ny,
                                  nscales=3, **PAR)
        gpu64.synchronize()
        got = flo.cpu().numpy()
        for k in range(G):
            assert np.array_equal(st[k].iterations(), want[k][2]), k
            assert np.array_equal(got[k], np.stack([want[k][0], want[k][1]], axis=-1).astype(np.float32)), k
    finally:
        gpu64.set_option("fuse3_afac1", 0)
        gpu64.set_option("fuse3_afac2", 0)


def linearised_state(orc, synth, nx, ny, seed=0):
    """A realistic inner-loop state: warp a synthetic pair with a small random flow."""
    I0, I1 = synth.pair_p1(nx, ny)
    rng = np.random.default_rng(seed)
    u1, u2 = rng.standard_normal((ny, nx)) * 0.5, rng.standard_normal((ny, nx)) * 0.5
    I1x, I1y = orc.centered_gradient(I1)
    I1w, I1wx, I1wy = (orc.bicubic_warp(x, u1, u2, True) for x in (I1, I1x, I1y))
    grad = I1wx * I1wx + I1wy * I1wy
    rho_c = I1w - I1wx * u1 - I1wy * u2 - I0
    p = [rng.standard_normal((ny, nx)) * 0.1 for _ in range(4)]
    return u1, u2, p, I1wx, I1wy, rho_c, grad


@pytest.mark.parametrize("nx,ny", [(5, 4), (62, 9), (63, 17), (64, 33), (125, 40), (200, 150), (640, 480)])
@pytest.mark.parametrize("n_iter", [1, 2, 7])
def test_iteration_kernel_bitexact(gpu64, orc, synth, nx, ny, n_iter):
    u1, u2, p, I1wx, I1wy, rho_c, grad = linearised_state(orc, synth, nx, ny)
    go = [x.copy() for x in (u1, u2, *p)]
    gg = [x.copy() for x in (u1, u2, *p)]
    e_o = orc.tvl1_iterations(*go, I1wx, I1wy, rho_c, grad, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
    e_g = gpu64.tvl1_iterations(*gg, I1wx, I1wy, rho_c, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
    for name, a, b in zip(("u1", "u2", "p11", "p12", "p21", "p22"), gg, go):
        assert np.array_equal(a, b), "%s differs: max %g" % (name, np.abs(a - b).max())
    assert abs(e_g - e_o) <= 1e-12 * max(abs(e_o), 1e-300)
    assert gpu64.stats().iterations()[0][0] == n_iter


@pytest.mark.parametrize("rows", [1, 3, 16, 64])
def test_iteration_kernel_any_strip_height(gpu64, orc, synth, rows):
    nx, ny = 130, 77
    u1, u2, p, I1wx, I1wy, rho_c, grad = linearised_state(orc, synth, nx, ny, seed=3)
    go = [x.copy() for x in (u1, u2, *p)]
    gg = [x.copy() for x in (u1, u2, *p)]
    orc.tvl1_iterations(*go, I1wx, I1wy, rho_c, grad, PAR["tau"], PAR["lam"], PAR["theta"], 5)
    gpu64.set_option("rows_per_wave", rows)
    try:
        gpu64.tvl1_iterations(*gg, I1wx, I1wy, rho_c, PAR["tau"], PAR["lam"], PAR["theta"], 5)
    finally:
        gpu64.set_option("rows_per_wave", 0)
    for a, b in zip(gg, go):
        assert np.array_equal(a, b)


def test_iteration_kernel_f32(gpu32, orc, synth):
    nx, ny = 200, 150
    u1, u2, p, I1wx, I1wy, rho_c, grad = linearised_state(orc, synth, nx, ny)
    go = [x.copy() for x in (u1, u2, *p)]
    gg = [x.copy() for x in (u1, u2, *p)]
    orc.tvl1_iterations(*go, I1wx, I1wy, rho_c, grad, PAR["tau"], PAR["lam"], PAR["theta"], 10)
    gpu32.tvl1_iterations(*gg, I1wx, I1wy, rho_c, PAR["tau"], PAR["lam"], PAR["theta"], 10)
    assert aepe(gg[0], gg[1], go[0], go[1]) < 1e-4
    for a, b in zip(gg[2:], go[2:]):
        assert np.abs(a - b).max() < 1e-4


@pytest.mark.parametrize("pair", ["P0", "P1"])
@pytest.mark.parametrize("nx,ny", [(64, 48), (135, 68)])
def test_single_scale(gpu64, orc, synth, pair, nx, ny):
    I0, I1 = synth.pair(pair, nx, ny)
    I0, I1 = orc.image_normalization_2(I0, I1)
    I0, I1 = orc.gaussian(I0, 0.8), orc.gaussian(I1, 0.8)
    z = np.zeros((ny, nx))
    uo, vo, it_o, err_o = orc.tvl1_single_scale(I0, I1, z, z, **PAR)
    ug, vg = gpu64.tvl1_single_scale(I0, I1, z, z, **PAR)
    st = gpu64.stats()
    assert list(st.iterations()[0]) == it_o
    assert np.allclose(st.errors()[0], err_o, rtol=1e-10, atol=0)
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9


@pytest.mark.parametrize("pair,nx,ny,nscales", [("P0", 64, 48, 3), ("P1", 135, 68, 3), ("P0", 640, 480, 5),
                                                ("P1", 640, 480, 5)])
def test_multiscale_matches_oracle(gpu64, orc, synth, pair, nx, ny, nscales):
    I0, I1 = synth.pair(pair, nx, ny)
    uo, vo, it_o, err_o = orc.tvl1_multiscale(I0, I1, nscales=nscales, **PAR)
    ug, vg = gpu64.tvl1_multiscale(I0, I1, nscales=nscales, **PAR)
    st = gpu64.stats()
    assert np.array_equal(st.iterations(), it_o)
    assert aepe(ug, vg, uo, vo) < 1e-4                    # the stated tolerance
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9   # what f64 storage actually achieves
    assert st.work_pix_iters == sum(int(it_o[s].sum()) * st.nx[s] * st.ny[s] for s in range(nscales))


def test_multiscale_f32_storage(gpu32, orc, synth):
    I0, I1 = synth.pair("P0", 640, 480)
    uo, vo, it_o, _ = orc.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    ug, vg = gpu32.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    assert aepe(ug, vg, uo, vo) < 1e-4


def test_known_answer_p0_640x480(gpu64, synth):
    """SURVEY.md §8c anchor measured on the compiled reference: mean(u,v) of the final flow."""
    I0, I1 = synth.pair("P0", 640, 480)
    u, v = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    assert abs(u.mean() - 1.594792) < 1e-6 and abs(v.mean() + 0.723327) < 1e-6
    assert list(gpu64.stats().iterations()[4]) == [26, 25, 5, 3, 3]


def test_fixed_work_runs_every_iteration(gpu64, synth):
    I0, I1 = synth.pair("P0", 128, 96)
    gpu64.set_option("fixed_work", 1)
    try:
        gpu64.tvl1_multiscale(I0, I1, nscales=2, warps=2, **PAR)
        assert (gpu64.stats().iterations() == 300).all()
    finally:
        gpu64.set_option("fixed_work", 0)


def test_sigma_too_large_is_reported(gpu64, ofx_mod, synth):
    I0, I1 = synth.pair("P0", 64, 48)
    with pytest.raises(ofx_mod.OfxError) as e:
        gpu64.tvl1_multiscale(I0, I1, nscales=6, **PAR)      # 64x48 -> ... -> 2x2: gaussian radius > size
    assert e.value.status == 2


# ---- BASELINE.json full sizes ---------------------------------------------------------------------------
def test_full_size_1080p_matches_oracle(gpu64, oracle_mod, synth):
    """config 2 at full size.  TV-L1 on the CPU is race-free, so the oracle may use every host core."""
    o = oracle_mod.Oracle()
    o.set_num_threads(min(oracle_mod.host_cores(), 32))
    I0, I1 = synth.pair("P1", 1920, 1080)
    try:
        uo, vo, it_o, _ = o.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    finally:
        o.set_num_threads(1)        # process-global (OpenMP): the SOR checks need the single-thread oracle
    ug, vg = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    assert np.array_equal(gpu64.stats().iterations(), it_o)
    assert aepe(ug, vg, uo, vo) < 1e-4
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9
    # the tolerance mode (what bench.py's headline runs) at the size the metric is quoted on: north_star's bar
    gpu64.set_option("relaxed_dual", 1)
    try:
        ut, vt = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
        it_t = gpu64.stats().iterations().copy()
    finally:
        gpu64.set_option("relaxed_dual", 0)
    assert aepe(ut, vt, uo, vo) < 1e-4 and aepe(ut, vt, uo, vo) < 1e-8      # stated tolerance; what it actually achieves
    assert np.array_equal(it_t, it_o)       # on the BASELINE configs the tolerance mode's iteration tables EQUAL the reference's


@pytest.mark.parametrize("pair,nx,ny", [("P0", 640, 480), ("P1", 640, 480), ("P0", 1920, 1080)])
def test_tolerance_mode_iteration_tables_equal_the_strict_ones_on_the_baseline_configs(gpu64, synth, pair, nx, ny):
    """BASELINE configs 1 and 2 (config 5's size: the 4K test below).  The tolerance mode changes the last bits of the dual update,
    which could move a stopping test across its threshold; on the configs the metric is quoted on it does not -- every one of the
    5 x 5 iteration counts equals the strict mode's, which is the reference's (README, bench.py ARITH say "equal": this
    is where that is asserted; the randomised soak allows +-4)."""
    I0, I1 = synth.pair(pair, nx, ny)
    us, vs = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    it_s = gpu64.stats().iterations().copy()
    gpu64.set_option("relaxed_dual", 1)
    try:
        ut, vt = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
        it_t = gpu64.stats().iterations().copy()
    finally:
        gpu64.set_option("relaxed_dual", 0)
    assert np.array_equal(it_t, it_s)
    assert aepe(ut, vt, us, vs) < 1e-8


@pytest.mark.timeout(900)
def test_full_size_4k_warps5_matches_oracle(gpu64, oracle_mod, synth):
    """BASELINE configs[4] size, the reference's parameters (warps=5): one 3840x2160 pair against the oracle on every
    host core (TV-L1 has no racy loop; ~30 s of CPU)."""
    o = oracle_mod.Oracle()
    o.set_num_threads(min(oracle_mod.host_cores(), 32))
    I0, I1 = synth.pair("P1", 3840, 2160, 1)
    try:
        uo, vo, it_o, _ = o.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    finally:
        o.set_num_threads(1)
    ug, vg = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    assert np.array_equal(gpu64.stats().iterations(), it_o)
    assert aepe(ug, vg, uo, vo) < 1e-4
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9
    gpu64.set_option("relaxed_dual", 1)     # config 5's size in the tolerance mode: tables equal, AEPE eight orders under the bar
    try:
        ut, vt = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
        it_t = gpu64.stats().iterations().copy()
    finally:
        gpu64.set_option("relaxed_dual", 0)
    assert np.array_equal(it_t, it_o)
    assert aepe(ut, vt, uo, vo) < 1e-8


def test_4k_size_independent_properties(gpu64, synth):
    """config 5 size (3840x2160), where the oracle is too slow for the suite: (i) the fused two-iteration
    kernel and the one-iteration kernel give bit-identical flows and iteration counts, for any strip
    height; (ii) the result is bit-reproducible run to run; (iii) zero motion stays zero."""
    I0, I1 = synth.pair("P1", 3840, 2160)
    kw = dict(nscales=5, warps=2, **PAR)
    u_a, v_a = gpu64.tvl1_multiscale(I0, I1, **kw)
    it_a = gpu64.stats().iterations()
    gpu64.set_option("fuse2", 0)
    try:
        u_b, v_b = gpu64.tvl1_multiscale(I0, I1, **kw)
        it_b = gpu64.stats().iterations()
    finally:
        gpu64.set_option("fuse2", 1)
    assert np.array_equal(it_a, it_b) and np.array_equal(u_a, u_b) and np.array_equal(v_a, v_b)
    gpu64.set_option("rows_per_wave2", 5)
    try:
        u_c, v_c = gpu64.tvl1_multiscale(I0, I1, **kw)
    finally:
        gpu64.set_option("rows_per_wave2", 0)
    assert np.array_equal(u_a, u_c) and np.array_equal(v_a, v_c)
    u_d, v_d = gpu64.tvl1_multiscale(I0, I1, **kw)
    assert np.array_equal(u_a, u_d) and np.array_equal(v_a, v_d)
    u_z, v_z = gpu64.tvl1_multiscale(I0, I0, **kw)
    assert np.abs(u_z).max() == 0.0 and np.abs(v_z).max() == 0.0
    assert np.isfinite(u_a).all() and abs(float(u_a.mean())) < 10


@pytest.mark.parametrize("nx,ny", [(61, 33), (121, 35), (180, 47), (64, 3), (3, 64)])
def test_fused_kernel_equals_single_kernel(gpu64, orc, synth, nx, ny):
    """strip / halo edge cases of the fused kernel: widths around multiples of 60, very flat / thin images"""
    u1, u2, p, I1wx, I1wy, rho_c, grad = linearised_state(orc, synth, max(nx, 8), max(ny, 8))
    u1, u2, I1wx, I1wy, rho_c, grad = (np.ascontiguousarray(a[:ny, :nx]) for a in (u1, u2, I1wx, I1wy, rho_c, grad))
    p = [np.ascontiguousarray(a[:ny, :nx]) for a in p]
    for n_iter in (2, 5, 6):
        go = [x.copy() for x in (u1, u2, *p)]
        gg = [x.copy() for x in (u1, u2, *p)]
        orc.tvl1_iterations(*go, I1wx, I1wy, rho_c, grad, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
        gpu64.tvl1_iterations(*gg, I1wx, I1wy, rho_c, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
        for a, b in zip(gg, go):
            assert np.array_equal(a, b)


# ---- lockstep groups (several pairs share every launch) -----------------------------------------------------
@pytest.mark.parametrize("G", [1, 2, 3, 5, 8, 16])
def test_lockstep_group_equals_pairs_solved_alone(ofx_mod, gpu64, orc, synth, G):
    """ofx_tvl1_group_dev: G different pairs through the same launches.  Every pair keeps its own stopping
    test / iteration counts / ping-pong phase: iteration tables equal the oracle's pair by pair and the .flo
    payloads equal the oracle's flow cast to float32 (bit for bit)."""
    import torch
    nx, ny = 150, 97
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                              [flo[k].data_ptr() for k in range(G)], nx, ny, nscales=3, **PAR)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    seen = set()
    for k in range(G):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], nscales=3, **PAR)
        assert np.array_equal(st[k].iterations(), it), k
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), k
        seen.add(tuple(int(x) for x in np.asarray(it).ravel()))
    if G >= 3:
        assert len(seen) > 1        # the pairs really stop at different iterations


def test_lockstep_group_f32_equals_single_pair_f32(ofx_mod, gpu32, synth):
    import torch
    nx, ny, G = 131, 80, 4
    pairs = [synth.pair("P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0].astype(np.float32)).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1].astype(np.float32)).cuda() for p in pairs]
    flo = torch.zeros((2, G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = gpu32.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                              [flo[0, k].data_ptr() for k in range(G)], nx, ny, nscales=3, **PAR)
    gpu32.synchronize()
    for k in range(G):
        gpu32.tvl1_multiscale_dev(d0[k].data_ptr(), d1[k].data_ptr(), flo[1, k].data_ptr(), nx, ny, nscales=3, **PAR)
        gpu32.synchronize()
        assert np.array_equal(gpu32.stats().iterations(), st[k].iterations())
    got = flo.cpu().numpy()
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("zfactor", [0.5, 0.62, 0.75, 0.9])
def test_fused_gaussian_kernels_agree(gpu64, synth, zfactor):
    """The pyramids' presmoothing / zoom Gaussians exist four times: two passes through memory (gauss_fused = 0), both passes in
    one launch for any radius (2), with the radius a template parameter and a register window in the column pass (3), and -- the
    default, 1 -- that kernel plus, for zfactor = 1/2, the whole zoom_out (smoothing + 2:1 sampling) in one launch (k_gauss_xy_dec;
    odd and even level sizes).  Same sums in the same order: the flows of a group must not differ by a bit."""
    import torch
    nx, ny, G = 203, 131, 3
    dev = torch.device("cuda")
    pairs = [synth.pair_device("P1", nx, ny, k, dev) for k in range(G)]
    outs = []
    for mode in (0, 1, 2, 3):
        gpu64.set_option("gauss_fused", mode)
        flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        gpu64.tvl1_group_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], [flo[k].data_ptr() for k in range(G)],
                             nx, ny, nscales=3, zfactor=zfactor, warps=2, **PAR)
        gpu64.synchronize()
        outs.append(flo)
    gpu64.set_option("gauss_fused", 1)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32))
    assert torch.equal(outs[0].view(torch.int32), outs[3].view(torch.int32))
    assert torch.isfinite(outs[1]).all() and float(outs[1].abs().max()) > 0.0


def test_lockstep_group_rejects_bad_sizes(ofx_mod, gpu64):
    with pytest.raises(ofx_mod.OfxError):
        gpu64.tvl1_group_dev([], [], [], 64, 48)
    with pytest.raises(ofx_mod.OfxError):
        gpu64.tvl1_group_dev([1] * 17, [1] * 17, [1] * 17, 64, 48)


def test_f32_mode_is_as_accurate_as_the_reference_own_float_build(gpu32, oracle_mod, synth):
    """SURVEY 8c, oracle variant 2: the reference compiled with ofpix_t = float (oracle/_ref/libofref32.so, its own
    sources through the include guard of src/of.h) drifts from its double build by AEPE ~1e-5 (TV-L1) / ~2e-6 (HS).
    The GPU's OFX_F32 mode (float storage, double arithmetic in registers, relaxed dual update) must stay in that band."""
    import os
    require_or_skip(oracle_mod.have_ref() and os.path.exists(oracle_mod.REF32_SO), "compiled reference (double + float builds) not present")
    r64, r32 = oracle_mod.Ref(), oracle_mod.Ref32()
    r64.set_num_threads(1)
    r32.set_num_threads(1)
    for pair in ("P0", "P1"):
        I0, I1 = synth.pair(pair, 320, 240)
        u, v = r64.tvl1_multiscale(I0, I1, nscales=4, **PAR)
        a, b = r32.tvl1_multiscale(I0, I1, nscales=4, **PAR)
        ug, vg = gpu32.tvl1_multiscale(I0, I1, nscales=4, **PAR)
        ref_drift, gpu_drift = aepe(a, b, u, v), aepe(ug, vg, u, v)
        assert gpu_drift < 1e-4                                  # the stated tolerance
        assert gpu_drift < 3 * ref_drift + 2e-6, (pair, gpu_drift, ref_drift)
        u, v = r64.hs_pyramidal(I0, I1, alpha=20.0, nscales=4, warps=5)
        a, b = r32.hs_pyramidal(I0, I1, alpha=20.0, nscales=4, warps=5)
        ug, vg = gpu32.hs_pyramidal(I0, I1, alpha=20.0, nscales=4, warps=5)
        ref_drift, gpu_drift = aepe(a, b, u, v), aepe(ug, vg, u, v)
        assert gpu_drift < 1e-3 and gpu_drift < 30 * ref_drift + 1e-5, (pair, gpu_drift, ref_drift)


@pytest.mark.parametrize("amp", [0.5, 6.0, 40.0])
def test_warp_tile_and_gather_paths_agree(gpu64, orc, synth, amp):
    """the LDS-staged warp (block bounding box of the taps fits the tile) and the global-gather path (it does not:
    amp = 40 scatters the samples of a block over +-40 pixels) compute the same bits; both against the oracle"""
    nx, ny = 200, 150
    I0, I1 = synth.pair("P1", nx, ny)
    rng = np.random.default_rng(7)
    u0 = rng.uniform(-amp, amp, (ny, nx))
    v0 = rng.uniform(-amp, amp, (ny, nx))
    uo, vo, it_o, _ = orc.tvl1_single_scale(I0, I1, u0, v0, warps=2, **PAR)
    res = {}
    for lds in (1, 0):
        gpu64.set_option("warp_lds", lds)
        try:
            res[lds] = gpu64.tvl1_single_scale(I0, I1, u0, v0, warps=2, **PAR)
            assert list(gpu64.stats().iterations()[0]) == list(it_o)
        finally:
            gpu64.set_option("warp_lds", 1)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    assert np.abs(res[1][0] - uo).max() < 1e-9 and np.abs(res[1][1] - vo).max() < 1e-9


@pytest.mark.parametrize("G", [1, 5])
def test_stored_intermediate_state_equals_recomputed_iteration(gpu64, orc, synth, G):
    """A loop that stops on the first iteration of a fused pair continues either from the intermediate state a
    predicting launch stored in the third buffer of the rotation (option store_a = 1 default, 2 = every launch) or
    from a recomputation of that iteration (store_a = 0): iteration tables and .flo payloads must be identical, and
    equal to the oracle's."""
    import torch
    nx, ny = 203, 131
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((3, G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    tables = []
    for mode in (0, 1, 2):
        gpu64.set_option("store_a", mode)
        try:
            st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                      [flo[mode, k].data_ptr() for k in range(G)], nx, ny, nscales=4, **PAR)
            gpu64.synchronize()
        finally:
            gpu64.set_option("store_a", 1)
        tables.append([s.iterations().copy() for s in st])
    got = flo.cpu().numpy()
    odd = 0
    for k in range(G):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], nscales=4, **PAR)
        for mode in (0, 1, 2):
            assert np.array_equal(tables[mode][k], it), (mode, k)
            assert np.array_equal(got[mode, k], np.stack([uo, vo], axis=-1).astype(np.float32)), (mode, k)
        odd += int((np.asarray(it) % 2 == 1).sum())
    assert odd > 0              # the case under test really occurs


# ---- the launch shapes bench.py times (round 3) ---------------------------------------------------------------------
# The fused kernel has a non-temporal-store instantiation (k_tvl1_iter2<T, true>) that the library selects once a launch's
# working set exceeds the Infinity Cache: 1080p groups of >= 2 pairs and every 4K launch, i.e. exactly the launches of the
# headline and of roofline_4k.  Small images never reach it on their own, so (i) the small group tests force it with option
# nt_stores = 1, for every store_a mode, and (ii) the timed shapes themselves are checked at full size.
@pytest.mark.parametrize("G", [1, 3, 5])
@pytest.mark.parametrize("store_a", [0, 1, 2])
def test_nt_store_kernel_in_small_groups_equals_oracle(gpu64, orc, synth, G, store_a):
    import torch
    nx, ny = 203, 131
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((2, G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    tables = {}
    for nt in (1, 2):                                      # 1 = always non-temporal, 2 = never
        gpu64.set_option("nt_stores", nt)
        gpu64.set_option("store_a", store_a)
        try:
            st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                      [flo[nt - 1, k].data_ptr() for k in range(G)], nx, ny, nscales=4, **PAR)
            gpu64.synchronize()
        finally:
            gpu64.set_option("nt_stores", 0)
            gpu64.set_option("store_a", 1)
        tables[nt] = [s.iterations().copy() for s in st]
    got = flo.cpu().numpy()
    assert np.array_equal(got[0], got[1])
    for k in range(G):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], nscales=4, **PAR)
        for nt in (1, 2):
            assert np.array_equal(tables[nt][k], it), (nt, k)
        assert np.array_equal(got[0, k], np.stack([uo, vo], axis=-1).astype(np.float32)), k


def _solo_flows(gpu, d0, d1, nx, ny, **kw):
    import torch
    out = torch.zeros((len(d0), ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    its = []
    for k in range(len(d0)):
        gpu.tvl1_multiscale_dev(d0[k].data_ptr(), d1[k].data_ptr(), out[k].data_ptr(), nx, ny, **kw)
        gpu.synchronize()
        its.append(gpu.stats().iterations().copy())
    return out, its


@pytest.mark.timeout(900)
def test_headline_launch_shape_1080p_group_of_5(gpu64, oracle_mod, synth):
    """bench.py's driver command (`--steps 20`): 1920x1080, lockstep groups of 5 pairs, non-temporal stores selected by the
    library itself.  Every payload equals the pair solved alone bit for bit, iteration tables too, and pair 0 equals the
    oracle (all host cores: TV-L1 has no racy loop)."""
    import torch
    nx, ny, G = 1920, 1080, 5
    kw = dict(nscales=5, warps=5, **PAR)
    host = [synth.pair("P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in host]
    d1 = [torch.from_numpy(p[1]).cuda() for p in host]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)],
                              nx, ny, **kw)
    gpu64.synchronize()
    solo, its = _solo_flows(gpu64, d0, d1, nx, ny, **kw)
    assert torch.equal(flo.view(torch.int32), solo.view(torch.int32))
    for k in range(G):
        assert np.array_equal(st[k].iterations(), its[k]), k
    o = oracle_mod.Oracle()
    o.set_num_threads(min(oracle_mod.host_cores(), 32))
    try:
        uo, vo, it_o, _ = o.tvl1_multiscale(host[0][0], host[0][1], nscales=5, **PAR)
    finally:
        o.set_num_threads(1)
    assert np.array_equal(st[0].iterations(), it_o)
    assert np.array_equal(flo[0].cpu().numpy(), np.stack([uo, vo], axis=-1).astype(np.float32))


@pytest.mark.timeout(900)
def test_headline_launch_shape_tolerance_mode_three_iterations_per_launch(gpu64, orc, synth):
    """What bench.py's headline times since round 3: 1920x1080, lockstep group of 5, the f64 tolerance mode, for which the
    library picks k_tvl1_iter3 (three iterations per launch, non-temporal stores) on the levels of >= 500 000 pixels x pairs.
    The pair solved alone runs the two-iteration kernel: every payload and iteration table must be equal bit for bit (same
    per-pixel functions; the stop inside a launch unit is finished by re-running the unit's first iterations), pair 0 stays
    within north_star's tolerance of the strict result -- and the 4K group of 4 (roofline_4k's launch) likewise."""
    import torch
    gpu64.set_option("relaxed_dual", 1)
    try:
        for nx, ny, G, warps in ((1920, 1080, 5, 5), (3840, 2160, 4, 2)):
            kw = dict(nscales=5, warps=warps, **PAR)
            dev = torch.device("cuda")
            pairs = [synth.pair_device("P1", nx, ny, k, dev) for k in range(G)]
            d0, d1 = [p[0] for p in pairs], [p[1] for p in pairs]
            flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            gpu64.set_option("fuse3", 2)
            st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)],
                                      nx, ny, **kw)
            gpu64.synchronize()
            gpu64.set_option("fuse3", 0)                       # the reference for the comparison: two iterations per launch
            solo, its = _solo_flows(gpu64, d0, d1, nx, ny, **kw)
            assert torch.equal(flo.view(torch.int32), solo.view(torch.int32)), (nx, ny)
            for k in range(G):
                assert np.array_equal(st[k].iterations(), its[k]), (nx, ny, k)
            del pairs, d0, d1, flo, solo
    finally:
        gpu64.set_option("relaxed_dual", 0)
        gpu64.set_option("fuse3", 2)


@pytest.mark.timeout(900)
def test_roofline_4k_launch_shape_group_of_4(gpu64, synth):
    """roofline_4k's launch (3840x2160, 4 pairs per launch, non-temporal stores): payloads and iteration tables equal the
    pairs solved alone (the solo 4K solve is checked against the oracle in test_full_size_4k_warps5_matches_oracle)."""
    import torch
    nx, ny, G = 3840, 2160, 4
    kw = dict(nscales=5, warps=2, **PAR)
    dev = torch.device("cuda")
    pairs = [synth.pair_device("P1", nx, ny, 1 + k, dev) for k in range(G)]
    d0, d1 = [p[0] for p in pairs], [p[1] for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)],
                              nx, ny, **kw)
    gpu64.synchronize()
    solo, its = _solo_flows(gpu64, d0, d1, nx, ny, **kw)
    assert torch.equal(flo.view(torch.int32), solo.view(torch.int32))
    for k in range(G):
        assert np.array_equal(st[k].iterations(), its[k]), k


@pytest.mark.timeout(900)
def test_batch_dev_on_four_contexts_at_1080p_equals_solo(ofx_mod, gpu64, synth):
    """ofx_tvl1_batch_dev as bench.py calls it: 10 1080p pairs (8 distinct, cycled) over 4 contexts -> groups of 3, 3, 3, 1
    on their own streams / host threads.  Payloads equal the pairs solved alone; the work record equals the solo tables."""
    import torch
    nx, ny, N = 1920, 1080, 10
    kw = dict(nscales=5, warps=5, **PAR)
    host = [synth.pair("P1", nx, ny, k) for k in range(8)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in host]
    d1 = [torch.from_numpy(p[1]).cuda() for p in host]
    flo = torch.zeros((N, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(4)]
    try:
        for c in ctxs:
            c.set_option("concurrency", 4)
        work = ofx_mod.tvl1_batch_dev(ctxs, [d0[i % 8].data_ptr() for i in range(N)], [d1[i % 8].data_ptr() for i in range(N)],
                                      [flo[i].data_ptr() for i in range(N)], nx, ny, **kw)
        for c in ctxs:
            c.synchronize()
    finally:
        for c in ctxs:
            c.close()
    solo, its = _solo_flows(gpu64, d0, d1, nx, ny, **kw)
    st = gpu64.stats()
    for i in range(N):
        assert torch.equal(flo[i].view(torch.int32), solo[i % 8].view(torch.int32)), i
        w = sum(int(its[i % 8][s].sum()) * st.nx[s] * st.ny[s] for s in range(5))
        assert work[i] == w, i


@pytest.mark.parametrize("pair", ["P0", "P1"])
def test_relaxed_dual_mode_stays_within_the_stated_tolerance(gpu64, orc, synth, pair):
    """Option relaxed_dual = 1 (the f64 "tolerance" mode: double storage, sqrt(x^2 + y^2) and one reciprocal per denominator in
    the dual update instead of the glibc-exact hypot and four IEEE divisions).  NOT bit-identical; the bar is north_star's:
    average end-point error against the reference < 1e-4 px (measured ~1e-8), iteration counts within a few per loop."""
    I0, I1 = synth.pair(pair, 640, 480)
    uo, vo, it_o, _ = orc.tvl1_multiscale(I0, I1, nscales=5, **PAR)
    gpu64.set_option("relaxed_dual", 1)
    try:
        ug, vg = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)
        it_g = gpu64.stats().iterations().copy()
    finally:
        gpu64.set_option("relaxed_dual", 0)
    assert aepe(ug, vg, uo, vo) < 1e-4                     # the stated tolerance
    assert np.abs(it_g - np.asarray(it_o)).max() <= 4
    us, vs = gpu64.tvl1_multiscale(I0, I1, nscales=5, **PAR)          # back in strict mode: bit-identical again
    assert np.array_equal(gpu64.stats().iterations(), it_o)
    assert np.abs(us - uo).max() < 1e-9 and np.abs(vs - vo).max() < 1e-9


@pytest.mark.parametrize("K", [4, 6])
@pytest.mark.parametrize("nx,ny", [(5, 4), (52, 4), (53, 5), (56, 8), (57, 9), (104, 17), (113, 16), (240, 135), (447, 301)])
def test_tile_kernel_iterations_bitexact(gpu64, orc, synth, K, nx, ny):
    """k_tvl1_tile (K = 4 / 6 iterations per launch, output tiles of (64 - 2 K) x (16 - 2 K) pixels): sizes around the tile
    pitch, iteration counts that end a launch unit early (1, K - 1, K, K + 1, 2 K + 3), against the oracle bit for bit."""
    u1, u2, p, I1wx, I1wy, rho_c, grad = linearised_state(orc, synth, max(nx, 8), max(ny, 8), seed=5)
    u1, u2, I1wx, I1wy, rho_c, grad = (np.ascontiguousarray(a[:ny, :nx]) for a in (u1, u2, I1wx, I1wy, rho_c, grad))
    p = [np.ascontiguousarray(a[:ny, :nx]) for a in p]
    gpu64.set_option("tile", K)
    gpu64.set_option("tile_max_px", 1e9)
    try:
        for n_iter in (1, K - 1, K, K + 1, 2 * K + 3):
            go = [x.copy() for x in (u1, u2, *p)]
            gg = [x.copy() for x in (u1, u2, *p)]
            e_o = orc.tvl1_iterations(*go, I1wx, I1wy, rho_c, grad, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
            e_g = gpu64.tvl1_iterations(*gg, I1wx, I1wy, rho_c, PAR["tau"], PAR["lam"], PAR["theta"], n_iter)
            for name, a, b in zip(("u1", "u2", "p11", "p12", "p21", "p22"), gg, go):
                assert np.array_equal(a, b), "%s differs after %d iterations: max %g" % (name, n_iter, np.abs(a - b).max())
            assert abs(e_g - e_o) <= 1e-12 * max(abs(e_o), 1e-300)
    finally:
        gpu64.set_option("tile", 0)
        gpu64.set_option("tile_max_px", 0)


@pytest.mark.parametrize("K", [4, 6])
@pytest.mark.parametrize("G", [1, 5, 16])
def test_tile_kernel_groups_stop_inside_launch_units(gpu64, orc, synth, K, G):
    """whole solves with the tile kernel at EVERY level (tile_max_px lifted): loops that end inside a launch unit are finished
    by the re-run of the unit's first iterations, every pair of a group with its own count; tables and .flo payloads
    against the oracle"""
    import torch
    nx, ny = 150, 97
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(q[0]).cuda() for q in pairs]
    d1 = [torch.from_numpy(q[1]).cuda() for q in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    gpu64.set_option("tile", K)
    gpu64.set_option("tile_max_px", 1e9)
    try:
        st = gpu64.tvl1_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                  [flo[k].data_ptr() for k in range(G)], nx, ny, nscales=3, **PAR)
        gpu64.synchronize()
    finally:
        gpu64.set_option("tile", 0)
        gpu64.set_option("tile_max_px", 0)
    got = flo.cpu().numpy()
    rem = set()
    for k in range(G):
        uo, vo, it, _ = orc.tvl1_multiscale(pairs[k][0], pairs[k][1], nscales=3, **PAR)
        assert np.array_equal(st[k].iterations(), it), k
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), k
        rem.update(int(x) % K for x in np.asarray(it).ravel())
    assert len(rem) > 1            # the loops really end at different positions inside a launch unit


def test_batch_dev_stages_images_that_are_not_on_the_contexts_gpu(ofx_mod, gpu64, synth):
    """ofx_tvl1_batch_dev accepts contexts on different GPUs (the C caller's multi-GPU path): a group whose images / payload
    arrays are not device memory of its context's GPU is staged through the context's arena and copied back.  On the one-GPU
    box the foreign memory is the host's -- pageable numpy arrays in, a pinned torch tensor and a pageable array out --; the
    payloads must equal the device-resident solves.  (Peer copies between GPUs take the same hipMemcpyDefault path;
    unmeasured on hardware here.)"""
    import torch
    nx, ny, N = 160, 120, 5
    kw = dict(nscales=3, warps=3, **PAR)
    host = [synth.pair("P1", nx, ny, k) for k in range(N)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in host]
    d1 = [torch.from_numpy(p[1]).cuda() for p in host]
    want, _ = _solo_flows(gpu64, d0, d1, nx, ny, **kw)
    pinned = torch.zeros((N, ny, nx, 2), dtype=torch.float32).pin_memory()
    pageable = np.zeros((N, ny, nx, 2), dtype=np.float32)
    h0 = [np.ascontiguousarray(p[0]) for p in host]
    h1 = [np.ascontiguousarray(p[1]) for p in host]
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(2)]
    try:
        for out_ptrs in ([pinned[k].data_ptr() for k in range(N)], [pageable[k].ctypes.data for k in range(N)]):
            # inputs: pairs 0, 1 pageable host arrays, pair 2 device-resident, pairs 3, 4 mixed
            i0 = [h0[0].ctypes.data, h0[1].ctypes.data, d0[2].data_ptr(), d0[3].data_ptr(), h0[4].ctypes.data]
            i1 = [h1[0].ctypes.data, h1[1].ctypes.data, d1[2].data_ptr(), h1[3].ctypes.data, d1[4].data_ptr()]
            ofx_mod.tvl1_batch_dev(ctxs, i0, i1, out_ptrs, nx, ny, **kw)
        assert np.array_equal(pinned.numpy().view(np.int32), want.cpu().numpy().view(np.int32))
        assert np.array_equal(pageable.view(np.int32), want.cpu().numpy().view(np.int32))
    finally:
        for c in ctxs:
            c.close()
