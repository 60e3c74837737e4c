"""Pins the oracle: our C restatement (oracle/ofx_oracle.c) against the reference's own compiled sources
(oracle/_ref/libofref.so) -- bit for bit, one OpenMP thread.  Skipped where /root/reference is absent
(the GPU box); there test_oracle_golden.py pins it against fixtures generated from the same build."""
import numpy as np
import pytest

SIZES = [(2, 2), (5, 7), (16, 16), (47, 33), (135, 68)]


def rnd(seed, ny, nx, scale=1.0, shift=0.0):
    return np.random.default_rng(seed).standard_normal((ny, nx)) * scale + shift


@pytest.mark.parametrize("nx,ny", SIZES)
def test_stencils(orc, ref, nx, ny):
    a, b = rnd(1, ny, nx), rnd(2, ny, nx)
    assert np.array_equal(orc.divergence(a, b), ref.divergence(a, b))
    for o, r in zip(orc.forward_gradient(a), ref.forward_gradient(a)):
        assert np.array_equal(o, r)
    for o, r in zip(orc.centered_gradient(a), ref.centered_gradient(a)):
        assert np.array_equal(o, r)
    for name in ("dxx", "dyy", "dxy"):
        assert np.array_equal(getattr(orc, name)(a), getattr(ref, name)(a)), name


@pytest.mark.parametrize("nx,ny", [(16, 16), (47, 33), (135, 68)])
def test_gaussian_bicubic_zoom_normalisation(orc, ref, nx, ny):
    a = rnd(3, ny, nx, 50, 100)
    for s in (0.8, 0.6 * np.sqrt(3.0), 1.3):
        assert np.array_equal(orc.gaussian(a, s), ref.gaussian(a, s))
    for f in (0.5, 0.7):
        assert np.array_equal(orc.zoom_out(a, f), ref.zoom_out(a, f))
    assert np.array_equal(orc.zoom_in(a, 2 * nx - 1, 2 * ny), ref.zoom_in(a, 2 * nx - 1, 2 * ny))
    u, v = rnd(5, ny, nx, 3), rnd(6, ny, nx, 3)
    for bo in (True, False):
        assert np.array_equal(orc.bicubic_warp(a, u, v, bo), ref.bicubic_warp(a, u, v, bo))
        assert np.array_equal(orc.bicubic_warp(a, 25 * u, 25 * v, bo), ref.bicubic_warp(a, 25 * u, 25 * v, bo))
    n1, n2 = orc.image_normalization_2(a, a * 0.5 + 3), ref.image_normalization_2(a, a * 0.5 + 3)
    assert np.array_equal(n1[0], n2[0]) and np.array_equal(n1[1], n2[1])
    assert orc.zoom_size(nx, ny, 0.5) == ref.zoom_size(nx, ny, 0.5)


def test_gaussian_throws_like_the_reference(orc, ref):
    a = rnd(3, 8, 4)
    with pytest.raises(ValueError):
        ref.gaussian(a, 0.8)            # operators.cpp:520-522
    with pytest.raises(ValueError):
        orc.gaussian(a, 0.8)


@pytest.mark.parametrize("pair,nx,ny,ns", [("P0", 64, 48, 3), ("P1", 135, 68, 3), ("P0", 160, 120, 4)])
def test_tvl1_multiscale(orc, ref, synth, pair, nx, ny, ns):
    I0, I1 = synth.pair(pair, nx, ny)
    uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=ns)
    ur, vr = ref.tvl1_multiscale(I0, I1, nscales=ns)
    assert np.array_equal(uo, ur) and np.array_equal(vo, vr)


def test_tvl1_single_scale_uses_incoming_flow(orc, ref, synth):
    I0, I1 = synth.pair("P1", 64, 48)
    u0, v0 = rnd(1, 48, 64, 0.3), rnd(2, 48, 64, 0.3)
    uo, vo, _, _ = orc.tvl1_single_scale(I0, I1, u0, v0, warps=3)
    ur, vr = ref.tvl1_single_scale(I0, I1, u0, v0, warps=3)
    assert np.array_equal(uo, ur) and np.array_equal(vo, vr)


def test_hs_and_brox(orc, ref, synth):
    I0, I1 = synth.pair("P1", 64, 48)
    uo, vo, _ = orc.hs_pyramidal(I0, I1, alpha=20.0, nscales=3, warps=5)
    ur, vr = ref.hs_pyramidal(I0, I1, alpha=20.0, nscales=3, warps=5)
    assert np.array_equal(uo, ur) and np.array_equal(vo, vr)
    uo, vo, _ = orc.brox_spatial(I0, I1, nscales=3, outer=5)
    ur, vr = ref.brox_spatial(I0, I1, nscales=3, outer=5)
    assert np.array_equal(uo, ur) and np.array_equal(vo, vr)


def test_brox_temporal(orc, ref, synth):
    """3 frames (no interior frame), 5 frames, and a case that runs into the 300-sweep limit"""
    for nx, ny, frames, kw in [(48, 40, 3, dict(nscales=2, outer=3)), (64, 48, 5, dict(nscales=2, outer=3, inner=2)),
                               (33, 47, 4, dict(nscales=1, outer=2, alpha=30.0, gamma=0.0))]:
        I = synth.sequence(nx, ny, frames)
        uo, vo, _ = orc.brox_temporal(I, **kw)
        ur, vr = ref.brox_temporal(I, **kw)
        assert np.array_equal(uo, ur) and np.array_equal(vo, vr)
    g = np.random.default_rng(0).standard_normal((4, 9, 7))
    for a, b in zip(orc.centered_gradient3(g), ref.centered_gradient3(g)):
        assert np.array_equal(a, b)
    assert np.array_equal(orc.image_normalization_1(g), ref.image_normalization_1(g))
    with pytest.raises(ValueError):
        orc.brox_temporal(g[:2])


def test_known_answer_anchor_p0_640x480(ref, synth):
    """mean(u, v) of the reference's flow on P0 640x480, 5 scales."""
    I0, I1 = synth.pair("P0", 640, 480)
    u, v = ref.tvl1_multiscale(I0, I1, nscales=5)
    assert abs(u.mean() - 1.594792) < 5e-7 and abs(v.mean() + 0.723327) < 5e-7


def test_colour_and_minmax_operators(orc, ref):
    """the remaining prototypes of SURVEY 8b's header ranges: bicubic_interpolation_at_color, getminmax"""
    rng = np.random.default_rng(11)
    img = rng.standard_normal((6, 9, 3)) * 40
    for uu, vv in [(-3.5, 2.5), (0.0, 0.0), (1.25, 3.5), (7.999, 4.2), (8.0, 5.0), (20.0, -9.0), (4.4, 2.6)]:
        for k in range(3):
            for bo in (False, True):
                assert orc.bicubic_at_color(img, uu, vv, k, bo) == ref.bicubic_at_color(img, uu, vv, k, bo)
    assert orc.getminmax(img) == ref.getminmax(img) == (img.min(), img.max())
