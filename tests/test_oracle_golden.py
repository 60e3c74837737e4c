"""Pins the oracle against the committed golden vectors (tests/golden/*.npz, generated from the compiled
reference by tests/golden/make_golden.py).  Runs everywhere, including the GPU box where the reference
does not exist.  Bit-exact: the oracle runs with one OpenMP thread like the generator did."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(G, "cases.json")))


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("tag", ["7x5", "16x16", "135x68"])
def test_operators(orc, tag):
    g = load("operators")
    a, b, img, u, v = (g["in_%s_%s" % (k, tag)] for k in ("a", "b", "img", "u", "v"))
    ny, nx = a.shape
    assert np.array_equal(orc.divergence(a, b), g["divergence_" + tag])
    fx, fy = orc.forward_gradient(a)
    assert np.array_equal(fx, g["fwd_x_" + tag]) and np.array_equal(fy, g["fwd_y_" + tag])
    cx, cy = orc.centered_gradient(a)
    assert np.array_equal(cx, g["cen_x_" + tag]) and np.array_equal(cy, g["cen_y_" + tag])
    for name in ("dxx", "dyy", "dxy"):
        assert np.array_equal(getattr(orc, name)(a), g[name + "_" + tag])
    assert np.array_equal(orc.bicubic_warp(img, u, v, True), g["warp_bo_" + tag])
    assert np.array_equal(orc.bicubic_warp(img, u, v, False), g["warp_nb_" + tag])
    n1, n2 = orc.image_normalization_2(img, img * 0.5 + 3)
    assert np.array_equal(n1, g["norm1_" + tag]) and np.array_equal(n2, g["norm2_" + tag])
    if "gauss08_" + tag in g:
        assert np.array_equal(orc.gaussian(img, 0.8), g["gauss08_" + tag])
        assert np.array_equal(orc.gaussian(img, 0.6 * np.sqrt(3.0)), g["gauss104_" + tag])
        assert np.array_equal(orc.zoom_out(img, 0.5), g["zoomout05_" + tag])
        assert np.array_equal(orc.zoom_out(img, 0.7), g["zoomout07_" + tag])
        assert np.array_equal(orc.zoom_in(a, 2 * nx - 1, 2 * ny), g["zoomin_" + tag])


def test_bicubic_at_and_zoom_size(orc):
    g = load("operators")
    img = g["in_img_7x5"]
    for bo, key in ((False, "at_nb"), (True, "at_bo")):
        got = np.array([orc.bicubic_at(img, x, y, bo) for x, y in g["at_points"]])
        assert np.array_equal(got, g[key])
    for nx, ny, nxx, nyy in g["zoom_sizes"][::2]:
        assert orc.zoom_size(int(nx), int(ny), 0.5) == (int(nxx), int(nyy))
    for nx, ny, nxx, nyy in g["zoom_sizes"][1::2]:
        assert orc.zoom_size(int(nx), int(ny), 0.75) == (int(nxx), int(nyy))


@pytest.mark.parametrize("case", sorted(CASES))
def test_solvers(orc, synth, case):
    c = CASES[case]
    g = load(case)
    if c["kind"] == "occ":          # TV-L1 with occlusions: "pair" is the variant of the three-frame synthetic sequence
        seq = synth.sequence(c["nx"], c["ny"], 3, c["pair"])
        u, v, chi, iters = orc.tvl1occ_multiscale(seq[0], seq[1], seq[2], **c["params"])
        assert np.array_equal(chi, g["chi"])
        out = (u, v, iters)
    elif c["kind"] == "broxt":      # temporal Brox: "pair" holds the number of frames of synth.sequence
        out = orc.brox_temporal(synth.sequence(c["nx"], c["ny"], c["pair"]), **c["params"])
    else:
        I0, I1 = synth.pair(c["pair"], c["nx"], c["ny"])
        fn = {"tvl1": orc.tvl1_multiscale, "hs": orc.hs_pyramidal, "brox": orc.brox_spatial, "rexpo": orc.robust_expo}[c["kind"]]
        out = fn(I0, I1, **c["params"])
    u, v, iters = out[0], out[1], out[2]
    assert np.array_equal(u, g["u"]) and np.array_equal(v, g["v"])
    # the reference prints coarse-to-fine; the oracle stores [scale][solve] with scale 0 = finest
    assert list(iters[::-1].ravel()) == list(g["iters"])


@pytest.mark.parametrize("tag", ["9x13", "24x19", "17x130"])
def test_occlusion_operators(orc, tag):
    """SURVEY 8f.1 / 8f.4 operators against the compiled reference's committed outputs (tests/golden/occ_operators.npz):
    Solver_wrt_v, Solver_wrt_chi (100 iterations from a zero dual variable), Solver_wrt_u (10 ROF iterations from zero dual
    planes), the ROF box sweep from given dual planes, medians, colour warp and the joint normalisations"""
    g = load("occ_operators")
    i = lambda k: g["in_%s_%s" % (k, tag)]
    u1, u2, chi = i("u1"), i("u2"), i("chi")
    args = (i("I1wx"), i("I1wy"), i("I_1wx"), i("I_1wy"), i("rho1_c"), i("rho3_c"))
    v = orc.occ_solver_v(u1, u2, chi, *args, i("grad1"), i("grad3"), 0.01, 0.3, 0.15)
    for k, a in zip(("v1", "v2", "vf1", "vf2", "vb1", "vb2"), v):
        assert np.array_equal(a, g["%s_%s" % (k, tag)]), k
    c100, _, _ = orc.occ_solver_chi(u1, u2, chi, *args, v[2], v[3], v[4], v[5], i("g"), 0.15, 0.3, 0.01, 0.15, 0.15, 0.15)
    assert np.array_equal(c100, g["chi100_" + tag])
    su1, su2, _ = orc.occ_solver_u(v[0], v[1], chi, i("g"), 0.3, 0.15)
    assert np.array_equal(su1, g["su1_" + tag]) and np.array_equal(su2, g["su2_" + tag])
    for a, k in zip(orc.rof_box(u1, i("roff"), i("P1"), i("P2"), i("g"), 0.3, 1.25, 3), ("rof_u", "rof_p1", "rof_p2")):
        assert np.array_equal(a, g["%s_%s" % (k, tag)]), k
    assert np.array_equal(orc.median_filtering(i("med"), 3), g["med3_" + tag])
    assert np.array_equal(orc.median_filtering(i("med"), 5), g["med5_" + tag])
    col = i("col")
    assert np.array_equal(orc.bicubic_warp_color(col, u1 * 3, u2 * 3, False), g["warpcol_nb_" + tag])
    assert np.array_equal(orc.bicubic_warp_color(col, u1 * 3, u2 * 3, True), g["warpcol_bo_" + tag])
    for a, k in zip(orc.image_normalization_2_color(col, col * 0.4 - 20), ("ncol1", "ncol2")):
        assert np.array_equal(a, g["%s_%s" % (k, tag)])
    for a, k in zip(orc.image_normalization_4(i("I1wx"), i("I1wy") * 2 + 30, i("rho1_c"), chi), "abcd"):
        assert np.array_equal(a, g["n4%s_%s" % (k, tag)])
    for a, k in zip(orc.image_normalization_3(i("I1wx"), i("I1wy") * 2 + 30, i("rho1_c")), "abc"):
        assert np.array_equal(a, g["n3%s_%s" % (k, tag)])


@pytest.mark.parametrize("batch", [1, 7, 64])
def test_hyperplane_schedule_is_exact(orc, synth, batch):
    """oracle order 2 = the HIP path's exact SOR schedule (pixel X of sweep s at time pos(X) + C s, many
    sweeps in flight, rollback + redo of an overshooting batch) run on the CPU: bit-identical to the
    reference's sequential sweeps for any batch size."""
    import ctypes as C
    fn = orc._fn("set_plane_batch", None, C.c_int)
    for kind, kw in (("hs", dict(alpha=20.0, nscales=3, warps=4)), ("brox", dict(nscales=3, outer=4))):
        I0, I1 = synth.pair("P1", 96, 64)
        run = orc.hs_pyramidal if kind == "hs" else orc.brox_spatial
        u0, v0, it0 = run(I0, I1, **kw)
        fn(batch)
        orc.set_sor_order(2)
        try:
            u2, v2, it2 = run(I0, I1, **kw)
        finally:
            orc.set_sor_order(0)
            fn(64)
        assert np.array_equal(it0, it2) and np.array_equal(u0, u2) and np.array_equal(v0, v2)
