"""Link compatibility: tests/shim/shim_driver.cpp is a caller written against the reference's own headers and
function names; its definitions come from include/ofx_reference_shim.hpp + libofx.so.  Built in the build container
(tests/shim/Makefile, needs /root/reference for the headers), run here on the GPU and compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import require_or_skip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "shim", "_build", "shim_driver")


def test_reference_prototypes_through_the_shim(orc, synth, tmp_path):
    require_or_skip(os.path.exists(DRIVER), "tests/shim/_build/shim_driver not built (needs /root/reference headers at build time)")
    nx, ny, frames = 96, 64, 4
    seq = synth.sequence(nx, ny, frames)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<iii", nx, ny, frames))
        f.write(seq.astype(np.float64).tobytes())
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    pos = [0]

    def take(shape):
        n = int(np.prod(shape))
        a = out[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return a
    I0, I1, S = seq[0], seq[1], (ny, nx)
    assert np.array_equal(take(S), orc.divergence(I0, I1))
    for want in orc.forward_gradient(I0) + orc.centered_gradient(I0):
        assert np.array_equal(take(S), want)
    for want in (orc.dxx(I0), orc.dyy(I0), orc.dxy(I0), orc.gaussian(I0, 0.8)):
        assert np.array_equal(take(S), want)
    i, j = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    k = i * nx + j
    u, v = 1.5 + 0.01 * (k % nx), -0.75 + 0.02 * (k // nx)
    assert np.array_equal(take(S), orc.bicubic_warp(I1, u, v, True))
    at = take((2,))
    assert at[0] == orc.bicubic_at(I0, 3.25, 2.5, False) and at[1] == orc.bicubic_at(I0, -1.0, 2.5, True)
    z = orc.zoom_out(I0, 0.5)
    assert np.array_equal(take(z.shape), z)
    assert np.array_equal(take(S), orc.zoom_in(z, nx, ny))
    for want in orc.image_normalization_2(I0, I1):
        assert np.array_equal(take(S), want)
    mm = take((2,))
    assert (mm[0], mm[1]) == (I0.min(), I0.max())
    uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=3)
    assert np.abs(take(S) - uo).max() < 1e-9 and np.abs(take(S) - vo).max() < 1e-9
    uo, vo, _ = orc.hs_pyramidal(I0, I1, alpha=20.0, nscales=3, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
    assert np.abs(take(S) - uo).max() < 1e-12 and np.abs(take(S) - vo).max() < 1e-12
    uo, vo, _ = orc.brox_spatial(I0, I1, alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=1, outer=4)
    assert np.abs(take(S) - uo).max() < 1e-11 and np.abs(take(S) - vo).max() < 1e-11
    uo, vo = orc.hs_classic(I0, I1, 25, 15.0)
    assert np.array_equal(take(S), uo) and np.array_equal(take(S), vo)
    uo, vo, _ = orc.brox_temporal(seq, alpha=18.0, gamma=7.0, nscales=2, nu=0.75, TOL=1e-4, inner=1, outer=3)
    T = (frames - 1, ny, nx)
    assert np.abs(take(T) - uo).max() < 1e-11 and np.abs(take(T) - vo).max() < 1e-11
    uo, vo, co, _ = orc.tvl1occ_multiscale(seq[0], seq[1], seq[2], filtI0=seq[1], nscales=3, warps=2)
    assert np.array_equal(take(S), uo) and np.array_equal(take(S), vo) and np.array_equal(take(S), co)
    assert np.array_equal(take(S), orc.median_filtering(seq[0], 3))
    for want in orc.image_normalization_4(seq[0], seq[1], seq[2], seq[1]):
        assert np.array_equal(take(S), want)
    uo, vo, _ = orc.robust_expo(I0, I1, method=2, alpha=18.7, gamma=5.0, lam=0.05, nscales=2, nu=0.5, TOL=1e-4, inner=1, outer=3)
    assert np.abs(take(S) - uo).max() < 1e-11 and np.abs(take(S) - vo).max() < 1e-11
    assert pos[0] == out.size
