"""GPU parity of the operators next to the hot path: colour warp and normalisations, the median
filter and the two pointwise / stencil solvers of TV-L1 with occlusions -- bit for bit against the oracle, which
tests/test_oracle_vs_ref.py pins against the compiled reference (Solver_wrt_chi: from a zero dual variable)."""
import numpy as np
import pytest

from test_oracle_vs_ref import _occ_inputs

pytestmark = pytest.mark.gpu


# Two schedules exist for the two iterative solvers inside the outer loop, both bit-identical by construction and both tested:
# * ROF box sweeps: all iterations of a call in flight (option rof_pipe = 1, default: sweeps ROF_LAGI positions apart, alfa of
#   the next iteration computed between the wavefronts) or one iteration at a time (0);
# * Solver_wrt_chi: CHI_N iterations per launch on overlapping LDS tiles (chi_fuse = 1, default) or two launches per iteration (0).
# * window length of the sweeps (rof_window): 10 steps per launch (default; two LDS windows per CU), or the round-2 geometry
#   of 24 steps in the "...24" runs.
@pytest.fixture(autouse=True, params=[(1, 0), (1, 24), (0, 0), (0, 24)], ids=["pipelined", "pipelined24", "serial", "serial24"])
def rof_schedule(request, gpu64):
    pipe, window = request.param
    if (pipe, window) != (1, 0) and not request.node.name.startswith(("test_rof", "test_occlusion_solvers", "test_tvl1occ_multiscale",
                                                                      "test_tvl1occ_lockstep")):
        pytest.skip("does not reach the iterative solvers: once is enough")
    gpu64.set_option("rof_pipe", pipe)
    gpu64.set_option("chi_fuse", pipe)
    gpu64.set_option("rof_window", window)
    yield request.param
    gpu64.set_option("rof_pipe", 1)
    gpu64.set_option("chi_fuse", 1)
    gpu64.set_option("rof_window", 0)


@pytest.mark.parametrize("ny,nx,nz", [(9, 13, 3), (40, 70, 3), (17, 8, 2), (6, 6, 1), (130, 97, 4)])
def test_colour_warp_and_normalisation_bitexact(gpu64, orc, ny, nx, nz):
    rng = np.random.default_rng(nx * ny + nz)
    I = rng.random((ny, nx, nz)) * 255
    u, v = rng.standard_normal((ny, nx)) * 3, rng.standard_normal((ny, nx)) * 3
    u[0, :] = -30.0                                    # far out of range
    for bo in (False, True):
        assert np.array_equal(gpu64.bicubic_warp_color(I, u, v, bo), orc.bicubic_warp_color(I, u, v, bo))
    J = rng.random((ny, nx, nz)) * 100 - 20
    for a, b in zip(gpu64.image_normalization_2_color(I, J), orc.image_normalization_2_color(I, J)):
        assert np.array_equal(a, b)
    K = np.full((ny, nx, nz), 7.0)
    K[..., 0] = I[..., 0]                              # one varying channel, the others constant: copy path per channel
    for a, b in zip(gpu64.image_normalization_2_color(K, K), orc.image_normalization_2_color(K, K)):
        assert np.array_equal(a, b)


def test_joint_normalisations_bitexact(gpu64, ofx_mod, orc):
    rng = np.random.default_rng(11)
    A, B, Cc, D = (rng.random((57, 43)) * s - o for s, o in ((255, 0), (90, 30), (300, 100), (10, 5)))
    for a, b in zip(gpu64.image_normalization_3(A, B, Cc), orc.image_normalization_3(A, B, Cc)):
        assert np.array_equal(a, b)
    for a, b in zip(gpu64.image_normalization_4(A, B, Cc, D), orc.image_normalization_4(A, B, Cc, D)):
        assert np.array_equal(a, b)
    Z = np.full((4, 5), 3.0)
    for a in gpu64.image_normalization_4(Z, Z, Z, Z):
        assert np.array_equal(a, Z)                    # max == min: copy
    with pytest.raises(ofx_mod.OfxError):
        gpu64.image_normalization_2_color(np.zeros((4, 4, 3)), np.zeros((4, 4, 3)), size=47)     # not a multiple of nz


@pytest.mark.parametrize("ny,nx", [(7, 9), (3, 3), (1, 6), (12, 2), (90, 131)])
def test_median_filter_bitexact(gpu64, ofx_mod, orc, ny, nx):
    rng = np.random.default_rng(ny * 100 + nx)
    I = np.round(rng.standard_normal((ny, nx)) * 4, 1)               # ties on purpose
    for w in (1, 3, 5):
        if (w >> 1) > min(nx, ny):
            with pytest.raises(ofx_mod.OfxError):
                gpu64.median_filtering(I, w)
            continue
        assert np.array_equal(gpu64.median_filtering(I, w), orc.median_filtering(I, w)), w
    with pytest.raises(ofx_mod.OfxError):
        gpu64.median_filtering(I, 11)


@pytest.mark.parametrize("nx,ny", [(19, 13), (8, 22), (150, 97)])
def test_occlusion_solvers_bitexact(gpu64, orc, nx, ny):
    rng = np.random.default_rng(nx + 1000 * ny)
    u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, grad1, grad3, g = _occ_inputs(rng, nx, ny)
    args_v = (u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, grad1, grad3, 0.01, 0.3, 0.15)
    vg, vo = gpu64.occ_solver_v(*args_v), orc.occ_solver_v(*args_v)
    for a, b in zip(vg, vo):
        assert np.array_equal(a, b)
    v1, v2, f1, f2, b1, b2 = vo
    par = (0.15, 0.3, 0.01, 0.15, 0.15, 0.15)                 # lambda, theta, alpha, beta, tau_chi, tau_eta
    args_c = (u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, f1, f2, b1, b2, g) + par
    cg, e1g, e2g = gpu64.occ_solver_chi(*args_c)              # 100 iterations from eta = 0, as the reference
    co, e1o, e2o = orc.occ_solver_chi(*args_c)
    assert np.array_equal(cg, co) and np.array_equal(e1g, e1o) and np.array_equal(e2g, e2o)
    # explicit state: 37 + 63 iterations with eta carried over = 100 iterations
    c1, a1, a2 = gpu64.occ_solver_chi(*args_c, n_iter=37)
    c2, _, _ = gpu64.occ_solver_chi(u1, u2, c1, *args_c[3:], eta1=a1, eta2=a2, n_iter=63)
    assert np.array_equal(c2, co)


@pytest.mark.parametrize("nx,ny", [(2, 2), (3, 2), (2, 5), (7, 6), (19, 13), (33, 140), (300, 131), (130, 260), (5, 400), (400, 3)])
def test_rof_box_and_solver_wrt_u_bitexact(gpu64, ofx_mod, orc, nx, ny):
    """the in-place box-relaxation sweep of Scalar_ROF_BoxCellCentered runs on hyperplanes of row blocks (24 steps per launch
    from an LDS copy of the launch window, blocks 32 steps apart): every dual value and u bit-identical to the sequential sweep,
    for all nine cell kinds, one and several row blocks (ny > 125, up to 4 here), images narrower than the lag between blocks,
    and Solver_wrt_u on top of it with the dual planes carried from call to call"""
    rng = np.random.default_rng(nx * 1000 + ny)
    u = rng.standard_normal((ny, nx))
    f = u / 0.3 + rng.standard_normal((ny, nx)) * 0.2
    P1, P2 = rng.standard_normal((ny, nx)) * 0.1, rng.standard_normal((ny, nx)) * 0.1
    g = 1.0 / (1.0 + rng.random((ny, nx)) * 3)
    for n_iter in (1, 3):
        for a, b in zip(gpu64.rof_box(u, f, P1, P2, g, 0.3, 1.25, n_iter), orc.rof_box(u, f, P1, P2, g, 0.3, 1.25, n_iter)):
            assert np.array_equal(a, b), (nx, ny, n_iter)
    v1, v2 = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    chi = np.clip(rng.random((ny, nx)) * 1.4 - 0.2, 0, 1)
    n_it = 10 if nx * ny < 20000 else 3
    g1, g2, gp = gpu64.occ_solver_u(v1, v2, chi, g, 0.3, 0.15, n_iter=n_it)          # from zero dual planes, as the reference
    o1, o2, op = orc.occ_solver_u(v1, v2, chi, g, 0.3, 0.15, n_iter=n_it)
    assert np.array_equal(g1, o1) and np.array_equal(g2, o2)
    for a, b in zip(gp, op):
        assert np.array_equal(a, b)
    g1, g2, _ = gpu64.occ_solver_u(v2, v1, chi, g, 0.3, 0.15, p=gp, n_iter=2)        # state carried over
    o1, o2, _ = orc.occ_solver_u(v2, v1, chi, g, 0.3, 0.15, p=op, n_iter=2)
    assert np.array_equal(g1, o1) and np.array_equal(g2, o2)
    if (nx, ny) == (2, 2):
        with pytest.raises(ofx_mod.OfxError):
            gpu64.rof_box(np.zeros((1, 5)), np.zeros((1, 5)), np.zeros((1, 5)), np.zeros((1, 5)), np.ones((1, 5)), 0.3, 1.25, 1)


def test_tvl1occ_multiscale(gpu64, synth, orc):
    """the whole TV-L1-with-occlusions solve, device resident, against the oracle (itself pinned against the compiled reference
    in tests/test_oracle_vs_ref.py): same outer-iteration table, flows and occlusion map bit-identical"""
    for nx, ny, ns, warps in ((64, 48, 2, 2), (90, 70, 3, 1), (160, 120, 3, 2), (200, 300, 4, 1)):
        seq = synth.sequence(nx, ny, 3, 1)
        kw = dict(lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=ns, zfactor=0.5, warps=warps, epsilon=0.01)
        uo, vo, co, it = orc.tvl1occ_multiscale(seq[0], seq[1], seq[2], **kw)
        u, v, c = gpu64.tvl1occ_multiscale(seq[0], seq[1], seq[2], **kw)
        st = gpu64.stats()
        got = np.array([[st.iters[s][w] for w in range(warps)] for s in range(ns)])
        assert np.array_equal(got, it), (nx, ny, got, it)
        assert np.array_equal(u, uo) and np.array_equal(v, vo) and np.array_equal(c, co), (nx, ny, np.abs(u - uo).max())


def test_tvl1occ_arguments(gpu64, ofx_mod):
    z = np.zeros((16, 16))
    for kw in (dict(nscales=0), dict(warps=0), dict(zfactor=1.0), dict(theta=0.0), dict(nscales=9)):
        with pytest.raises(ofx_mod.OfxError):
            gpu64.tvl1occ_multiscale(z, z, z, **kw)


def test_tvl1occ_batch_over_contexts(ofx_mod, gpu64, synth):
    """independent triples on several contexts (host thread + stream each) give what one context gives, in the order of the input"""
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(3)]
    kw = dict(nscales=3, warps=1)
    triples = []
    for k in range(5):
        seq = synth.sequence(96, 72, 3, k + 1)
        triples.append((seq[0], seq[1], seq[2]) if k % 2 else (seq[2], seq[1], seq[0], seq[1]))
    got = ofx_mod.tvl1occ_batch(ctxs, triples, **kw)
    for t, (u, v, c) in zip(triples, got):
        wu, wv, wc = gpu64.tvl1occ_multiscale(*t[:3], **kw)
        assert np.array_equal(u, wu) and np.array_equal(v, wv) and np.array_equal(c, wc)
    with pytest.raises(ofx_mod.OfxError):
        ofx_mod.tvl1occ_batch(ctxs, triples, nscales=0)


def test_tvl1occ_lockstep_group_with_different_iteration_counts(ofx_mod, gpu64, synth, orc):
    """one context, one lockstep group: triples whose outer loops end at different iterations (and at different levels) ride in
    the same launches, the finished ones frozen -- every result equal to the triple solved alone, and to the oracle"""
    ctx = ofx_mod.Ofx(0, ofx_mod.F64)
    kw = dict(nscales=3, warps=2, epsilon=0.002)
    triples = []
    for k in range(6):
        seq = synth.sequence(160, 120, 3, k + 1)
        triples.append((seq[0], seq[1], seq[2]) if k % 3 else (seq[2], seq[1], seq[1]))      # a static half for some: other counts
    got = ofx_mod.tvl1occ_batch([ctx], triples, **kw)
    tables = []
    for t, (u, v, c) in zip(triples, got):
        wu, wv, wc = gpu64.tvl1occ_multiscale(*t[:3], **kw)
        st = gpu64.stats()
        tables.append(tuple(st.iters[s][w] for s in range(3) for w in range(2)))
        assert np.array_equal(u, wu) and np.array_equal(v, wv) and np.array_equal(c, wc)
    assert len(set(tables)) > 1, tables                     # the group really was heterogeneous
    uo, vo, co, _ = orc.tvl1occ_multiscale(*triples[1][:3], **kw)
    assert np.array_equal(got[1][0], uo) and np.array_equal(got[1][2], co)
    ctx.set_option("lockstep", 4)                           # 4 + 2
    again = ofx_mod.tvl1occ_batch([ctx], triples, **kw)
    for a, b in zip(got, again):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
