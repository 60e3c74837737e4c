"""GPU parity: every operator of the C ABI against the oracle on the same seeded inputs.
f64 storage: bit-exact (the kernels keep the reference's IEEE association order, no FMA contraction).
f32 storage: inputs/outputs rounded to float -> relative tolerance 1e-5 of the data range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(2, 2), (5, 7), (16, 16), (47, 33), (135, 68), (640, 480)]


def rnd(seed, ny, nx, scale=1.0, shift=0.0):
    return np.random.default_rng(seed).standard_normal((ny, nx)) * scale + shift


@pytest.mark.parametrize("nx,ny", SIZES)
def test_stencils_bitexact(gpu64, orc, nx, ny):
    a, b = rnd(1, ny, nx), rnd(2, ny, nx)
    assert np.array_equal(gpu64.divergence(a, b), orc.divergence(a, b))
    for g, o in zip(gpu64.forward_gradient(a), orc.forward_gradient(a)):
        assert np.array_equal(g, o)
    for g, o in zip(gpu64.centered_gradient(a), orc.centered_gradient(a)):
        assert np.array_equal(g, o)
    assert np.array_equal(gpu64.dxx(a), orc.dxx(a))
    assert np.array_equal(gpu64.dyy(a), orc.dyy(a))
    assert np.array_equal(gpu64.dxy(a), orc.dxy(a))


@pytest.mark.parametrize("nx,ny", [(16, 16), (47, 33), (135, 68), (640, 480)])
@pytest.mark.parametrize("sigma", [0.8, 0.6 * np.sqrt(3.0), 1.7])
def test_gaussian_bitexact(gpu64, orc, nx, ny, sigma):
    a = rnd(3, ny, nx, 50, 100)
    assert np.array_equal(gpu64.gaussian(a, sigma), orc.gaussian(a, sigma))


def test_gaussian_sigma_too_large(gpu64, ofx_mod):
    a = rnd(3, 8, 4)
    with pytest.raises(ofx_mod.OfxError) as e:
        gpu64.gaussian(a, 0.8)          # radius 5 > width 4: the reference throws (operators.cpp:520-522)
    assert e.value.status == 2


@pytest.mark.parametrize("nx,ny", [(16, 16), (47, 33), (135, 68), (640, 480)])
def test_bicubic_and_zoom_bitexact(gpu64, orc, nx, ny):
    a = rnd(4, ny, nx, 50, 100)
    u, v = rnd(5, ny, nx, 3), rnd(6, ny, nx, 3)
    assert np.array_equal(gpu64.bicubic_warp(a, u, v, True), orc.bicubic_warp(a, u, v, True))
    assert np.array_equal(gpu64.bicubic_warp(a, u, v, False), orc.bicubic_warp(a, u, v, False))
    # far outside the image, negative coordinates, exact integers
    assert np.array_equal(gpu64.bicubic_warp(a, u * 30, v * 30, False), orc.bicubic_warp(a, u * 30, v * 30, False))
    assert np.array_equal(gpu64.bicubic_warp(a, np.round(u), np.round(v), True),
                          orc.bicubic_warp(a, np.round(u), np.round(v), True))
    assert np.array_equal(gpu64.zoom_out(a, 0.5), orc.zoom_out(a, 0.5))
    assert np.array_equal(gpu64.zoom_out(a, 0.7), orc.zoom_out(a, 0.7))
    assert np.array_equal(gpu64.zoom_in(a, 2 * nx - 1, 2 * ny), orc.zoom_in(a, 2 * nx - 1, 2 * ny))
    n1, n2 = gpu64.image_normalization_2(a, a * 0.5 + 3), orc.image_normalization_2(a, a * 0.5 + 3)
    assert np.array_equal(n1[0], n2[0]) and np.array_equal(n1[1], n2[1])
    c = np.full((ny, nx), 7.0)          # den == 0 -> plain copy (utils.cpp:318-325)
    n1 = gpu64.image_normalization_2(c, c)
    assert np.array_equal(n1[0], c) and np.array_equal(n1[1], c)


def test_bicubic_at_points(gpu64, orc):
    a = rnd(7, 20, 31, 50, 100)
    uu = np.array([-3.5, -0.5, 0.0, 0.25, 1.0, 1.5, 15.2, 27.999, 28.0, 29.5, 30.0, 30.5, 40.0])
    vv = np.array([2.5, -0.25, 0.0, 18.5, 1.0, 17.0, 3.3, 5.0, 19.0, 18.9, 21.0, 7.7, -9.0])
    for bo in (False, True):
        g = gpu64.bicubic_at(a, uu, vv, bo)
        o = np.array([orc.bicubic_at(a, x, y, bo) for x, y in zip(uu, vv)])
        assert np.array_equal(g, o)


@pytest.mark.parametrize("nx,ny", [(47, 33), (640, 480)])
def test_ops_f32_storage(gpu32, orc, nx, ny):
    a, b = rnd(1, ny, nx, 50, 100), rnd(2, ny, nx, 50, 100)
    tol = 1e-5 * 400
    assert np.abs(gpu32.divergence(a, b) - orc.divergence(a, b)).max() < tol
    assert np.abs(gpu32.gaussian(a, 0.8) - orc.gaussian(a, 0.8)).max() < tol
    u, v = rnd(5, ny, nx, 3), rnd(6, ny, nx, 3)
    assert np.abs(gpu32.bicubic_warp(a, u, v, True) - orc.bicubic_warp(a, u, v, True)).max() < 5e-3
    assert np.abs(gpu32.zoom_out(a, 0.5) - orc.zoom_out(a, 0.5)).max() < tol


def test_colour_sequence_and_minmax_operators(gpu64, orc, ofx_mod):
    """ofx_bicubic_at_color, ofx_centered_gradient3, ofx_image_normalization_1, ofx_getminmax, ofx_zoom_out_color
    (the prototypes of SURVEY 8b's header ranges beyond the single-channel ones)"""
    rng = np.random.default_rng(12)
    img = rng.standard_normal((13, 17, 3)) * 40
    uu, vv = rng.uniform(-3, 20, 200), rng.uniform(-3, 16, 200)
    for k in range(3):
        for bo in (False, True):
            want = np.array([orc.bicubic_at_color(img, x, y, k, bo) for x, y in zip(uu, vv)])
            assert np.array_equal(gpu64.bicubic_at_color(img, uu, vv, k, bo), want)
    for nz in (1, 2, 5):
        f = rng.standard_normal((nz, 11, 14))
        for g, o in zip(gpu64.centered_gradient3(f), orc.centered_gradient3(f)):
            assert np.array_equal(g, o)
    seq = np.floor(rng.uniform(3, 200, (4, 9, 12)))
    assert np.array_equal(gpu64.image_normalization_1(seq), orc.image_normalization_1(seq))
    const = np.full((3, 4), 7.0)
    assert np.array_equal(gpu64.image_normalization_1(const), const)          # den = 0: plain copy (utils.cpp:269-274)
    assert gpu64.getminmax(img) == (img.min(), img.max())
    one = rng.uniform(0, 255, (20, 24, 1))
    assert np.array_equal(gpu64.zoom_out_color(one, 0.5)[..., 0], orc.zoom_out(one[..., 0], 0.5))
    with pytest.raises(ofx_mod.OfxError):
        gpu64.zoom_out_color(img, 0.5)                                        # nz > 1: undefined in the reference


def test_hypot_is_glibc_hypot_over_the_whole_range(gpu64):
    """ofx_hypot = the device's restatement of glibc's hypot (src/tvl1flow.cpp:172-173 calls libm).  The common domain
    (ax <= 2^511, ay >= 2^-383) runs the compiler's sqrt / division expansions without their range scaling; everything else
    the general code.  Both against the libm of this box (numpy.hypot calls it), bit for bit: magnitudes from denormal to
    1e300, the domain borders, ratios around 2^54, zeros, exact triples, equal arguments."""
    rng = np.random.default_rng(11)
    n = 400000
    xs = [rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 3, n), rng.standard_normal(n) * 10.0 ** rng.uniform(-320, 305, n)]
    ys = [rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 3, n), rng.standard_normal(n) * 10.0 ** rng.uniform(-320, 305, n)]
    # around the borders of the common domain and of the general code's own branches
    for e in (-1074, -1022, -600, -460, -459, -458, -384, -383, -382, -54, 0, 54, 510, 511, 512, 600, 1023):
        m = 1.0 + rng.random(4000)
        xs.append(np.ldexp(m, e))
        ys.append(np.ldexp(1.0 + rng.random(4000), e + rng.integers(-60, 2, 4000)))
    xs.append(np.array([0.0, -0.0, 3.0, 5.0, 8.0, 1e-200, 2.5, 0.0, 1.0, 1e308]))
    ys.append(np.array([0.0, 7.5, 4.0, 12.0, 15.0, 0.0, 2.5, -0.0, 2.0 ** -54, 1e308]))
    x, y = np.concatenate(xs), np.concatenate(ys)
    with np.errstate(over="ignore"):
        want = np.hypot(x, y)
    got = gpu64.hypot(x, y)
    bad = np.flatnonzero(got.view(np.int64) != want.view(np.int64))
    assert bad.size == 0, (bad.size, x[bad[:5]], y[bad[:5]], got[bad[:5]], want[bad[:5]])
