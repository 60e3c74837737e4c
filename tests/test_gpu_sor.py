"""GPU parity of the Horn-Schunck-pyramidal and Brox-spatial paths.

The reference sweeps its in-place SOR in a fixed sequential order (interior rows lexicographic, then border
rows, border columns, corners).  The HIP path has two modes (option "sor_exact"):

 * exact (default): the same visiting order, pipelined over hyperplanes with many sweeps in flight
   (ofx_sor.hip).  Every pixel reads exactly the versions the sequential sweep reads, so
   sweep counts equal the reference's and the flow is bit-identical up to the summation order of the
   stopping criterion -- asserted against the oracle in the REFERENCE order at < 1e-12.
 * colour order (sor_exact = 0, fast): 4-colour / red-black sweeps.  Kernels are checked bit-for-bit
   against oracle.set_sor_order(1); versus the reference order the result moves by AEPE 1e-5 .. 6e-4
   depending on the input, which is why it is not the default.
"""
import numpy as np
import pytest
from conftest import aepe

pytestmark = pytest.mark.gpu

# The windowed exact sweeps exist twice: with one global round trip per time step (k_hs_window / k_brox_window, what lockstep
# groups of >= 4 pairs use) and with the launch window staged in LDS (k_*_window_lds, what a lone solve uses).  Every test
# runs with both forced (option sor_lds = 0 / 2), the full-size configurations with the default choice only.
SOR_DEFAULT_ONLY = ("test_cfg3", "test_cfg4", "test_hs_classic")


@pytest.fixture(autouse=True, params=[0, 2], ids=["global", "lds"])
def sor_kernel(request, gpu64, gpu32):
    if request.node.name.startswith(SOR_DEFAULT_ONLY):
        if request.param == 0:
            pytest.skip("full-size case: default kernel choice only")
        yield 1
        return
    for c in (gpu64, gpu32):
        c.set_option("sor_lds", request.param)
    yield request.param
    for c in (gpu64, gpu32):
        c.set_option("sor_lds", 1)


@pytest.fixture()
def colour_orc(orc, gpu64):
    """oracle AND GPU context both in colour-order mode"""
    orc.set_sor_order(1)
    gpu64.set_option("sor_exact", 0)
    gpu64.set_option("sor_fuse", -1)        # the one-launch-per-colour kernels; the tile sweeps: tests/test_gpu_sor_tile.py
    yield orc
    orc.set_sor_order(0)
    gpu64.set_option("sor_exact", 1)
    gpu64.set_option("sor_fuse", 0)


@pytest.mark.parametrize("pair,nx,ny,warps", [("P0", 64, 48, 4), ("P1", 135, 68, 3), ("P1", 33, 47, 4), ("P1", 9, 8, 2)])
def test_hs_single_scale_exact(gpu64, orc, synth, pair, nx, ny, warps):
    I1, I2 = synth.pair(pair, nx, ny)
    z = np.zeros((ny, nx))
    uo, vo, it_o = orc.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=warps)         # reference order
    ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=warps)
    assert list(gpu64.stats().iterations()[0]) == it_o
    assert np.abs(ug - uo).max() < 1e-12 and np.abs(vg - vo).max() < 1e-12


@pytest.mark.parametrize("batch", [1, 5, 64, 300])
def test_hs_exact_any_batch_size(gpu64, orc, synth, batch):
    """sweeps in flight per batch must not matter (rollback + redo of the overshooting batch)"""
    I1, I2 = synth.pair("P1", 80, 50)
    z = np.zeros((50, 80))
    uo, vo, it_o = orc.hs_single_scale(I1, I2, z, z, alpha=10.0, warps=3)
    gpu64.set_option("sor_batch", batch)
    try:
        ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=10.0, warps=3)
    finally:
        gpu64.set_option("sor_batch", 0)
    assert list(gpu64.stats().iterations()[0]) == it_o
    assert np.abs(ug - uo).max() < 1e-12 and np.abs(vg - vo).max() < 1e-12


@pytest.mark.parametrize("pair,nx,ny,ns", [("P0", 64, 48, 3), ("P1", 320, 240, 4)])
def test_hs_pyramidal_exact(gpu64, orc, synth, pair, nx, ny, ns):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=20.0, nscales=ns, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    ur, vr, it_r = orc.hs_pyramidal(I1, I2, **kw)
    ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert aepe(ug, vg, ur, vr) < 1e-4                                        # the stated tolerance
    assert np.abs(ug - ur).max() < 1e-12 and np.abs(vg - vr).max() < 1e-12    # what the exact schedule achieves


@pytest.mark.parametrize("pair,nx,ny,ns", [("P0", 64, 48, 3), ("P1", 160, 120, 3), ("P0", 320, 240, 4)])
def test_brox_exact(gpu64, orc, synth, pair, nx, ny, ns):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=ns, nu=0.5, TOL=1e-4, inner=1, outer=6)
    ur, vr, it_r = orc.brox_spatial(I1, I2, **kw)
    ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert aepe(ug, vg, ur, vr) < 1e-4
    assert np.abs(ug - ur).max() < 1e-11 and np.abs(vg - vr).max() < 1e-11


# ---- BASELINE.json configs 3 and 4 at full size, against the single-thread oracle (its only deterministic mode) ------
@pytest.mark.timeout(600)
def test_cfg3_hs_1080p_full_size_matches_oracle(gpu64, orc, synth):
    """BASELINE configs[2]: horn_schunck_pyramidal 1920x1080, alpha=20 nscales=5 warps=10 (~4-9 s of CPU).
    SURVEY 8c anchor: 1004 sweeps, mean(u, v) = (1.556167, -0.751293) on P0."""
    I1, I2 = synth.pair("P0", 1920, 1080)
    kw = dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    ur, vr, it_r = orc.hs_pyramidal(I1, I2, **kw)
    ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert int(np.asarray(it_r).sum()) == 1004
    assert abs(ug.mean() - 1.556167) < 1e-6 and abs(vg.mean() + 0.751293) < 1e-6
    assert aepe(ug, vg, ur, vr) < 1e-4
    assert np.abs(ug - ur).max() < 1e-12 and np.abs(vg - vr).max() < 1e-12


@pytest.mark.timeout(600)
def test_cfg4_brox_720p_full_size_matches_oracle(gpu64, orc, synth):
    """BASELINE configs[3]: brox_optic_flow_spatial 1280x720, default alpha / gamma -> 6 scales, inner 1, outer 15
    (~3-8 s of CPU).  SURVEY 8c anchor: 1764 sweeps, mean(u, v) = (1.526222, -0.762388) on P0."""
    I1, I2 = synth.pair("P0", 1280, 720)
    kw = dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)
    ur, vr, it_r = orc.brox_spatial(I1, I2, **kw)
    ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert int(np.asarray(it_r).sum()) == 1764
    assert abs(ug.mean() - 1.526222) < 1e-6 and abs(vg.mean() + 0.762388) < 1e-6
    assert aepe(ug, vg, ur, vr) < 1e-4
    assert np.abs(ug - ur).max() < 1e-12 and np.abs(vg - vr).max() < 1e-12


@pytest.mark.parametrize("pair,nx,ny", [("P0", 64, 48), ("P1", 135, 68), ("P1", 33, 47)])
def test_hs_single_scale_same_order(gpu64, colour_orc, synth, pair, nx, ny):
    I1, I2 = synth.pair(pair, nx, ny)
    I1, I2 = colour_orc.image_normalization_2(I1, I2)
    I1, I2 = colour_orc.gaussian(I1, 0.8), colour_orc.gaussian(I2, 0.8)
    z = np.zeros((ny, nx))
    uo, vo, it_o = colour_orc.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=4)
    ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=4)
    assert list(gpu64.stats().iterations()[0]) == it_o
    assert np.abs(ug - uo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-9


@pytest.mark.parametrize("pair,nx,ny,ns", [("P0", 64, 48, 3), ("P1", 320, 240, 4)])
def test_hs_pyramidal_colour_order(gpu64, colour_orc, synth, pair, nx, ny, ns):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=20.0, nscales=ns, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    uc, vc, it_c = colour_orc.hs_pyramidal(I1, I2, **kw)
    ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_c)
    assert np.abs(ug - uc).max() < 1e-9 and np.abs(vg - vc).max() < 1e-9


def test_hs_maxiter_and_tol_edge_cases(gpu64, colour_orc, synth):
    I1, I2 = synth.pair("P0", 64, 48)
    z = np.zeros((48, 64))
    for kw in (dict(maxiter=0), dict(maxiter=3), dict(TOL=2000.0)):
        uo, vo, it_o = colour_orc.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=2, **kw)
        ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=2, **kw)
        assert list(gpu64.stats().iterations()[0]) == it_o
        assert np.abs(ug - uo).max() < 1e-9


@pytest.mark.parametrize("pair,nx,ny,ns", [("P0", 64, 48, 3), ("P1", 160, 120, 3)])
def test_brox_same_order(gpu64, colour_orc, synth, pair, nx, ny, ns):
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=ns, nu=0.5, TOL=1e-4, inner=2, outer=5)
    uo, vo, it_o = colour_orc.brox_spatial(I1, I2, **kw)
    ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_o)
    assert np.abs(ug - uo).max() < 1e-8 and np.abs(vg - vo).max() < 1e-8


def test_colour_order_drift_is_bounded(gpu64, orc, synth):
    """documents why colour order is not the default: result drifts from the reference order"""
    nx, ny = 160, 120
    I1, I2 = synth.pair("P0", nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=1, outer=8)
    ur, vr, _ = orc.brox_spatial(I1, I2, **kw)
    gpu64.set_option("sor_exact", 0)
    try:
        ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    finally:
        gpu64.set_option("sor_exact", 1)
    assert aepe(ug, vg, ur, vr) < 1e-3


def test_exact_mode_tiny_image_falls_back(gpu64, colour_orc, synth):
    """nx or ny < 3 has no interior: the exact schedule is not defined, the colour sweep is used"""
    I1, I2 = synth.pair("P1", 7, 2)
    z = np.zeros((2, 7))
    gpu64.set_option("sor_exact", 1)
    uo, vo, it_o = colour_orc.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=2)
    ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=2)
    assert list(gpu64.stats().iterations()[0]) == it_o and np.abs(ug - uo).max() < 1e-9


def test_sor_f32_storage(gpu32, orc, synth):
    I1, I2 = synth.pair("P1", 160, 120)
    ur, vr, _ = orc.hs_pyramidal(I1, I2, alpha=20.0, nscales=3, warps=5)
    ug, vg = gpu32.hs_pyramidal(I1, I2, alpha=20.0, nscales=3, warps=5)
    assert aepe(ug, vg, ur, vr) < 1e-3


# ---- the two implementations of the exact schedule ----------------------------------------------------------
@pytest.mark.parametrize("mode,window,rows", [(2, 0, 0), (1, 1, 2), (1, 5, 3), (1, 8, 7), (1, 32, 0), (1, 200, 16), (1, 8, 1000),
                                              (1, 4, 5), (1, 16, 0), (1, 16, 9), (1, 24, 6), (1, 8, 125), (1, 16, 253)])
def test_exact_schedule_variants_agree_with_reference(gpu64, orc, synth, mode, window, rows):
    """sor_exact = 2: one launch per time step; sor_exact = 1: K time steps per launch (windowed, one workgroup
    per (sweep, row block), snapshots instead of rollback).  Both must reproduce the reference order for any window
    length and block height."""
    I1, I2 = synth.pair("P1", 90, 61)
    z = np.zeros((61, 90))
    uo, vo, it_o = orc.hs_single_scale(I1, I2, z, z, alpha=12.0, warps=3)
    kw = dict(alpha=50.0, gamma=10.0, nscales=2, nu=0.5, TOL=1e-4, inner=2, outer=3)
    ur, vr, it_r = orc.brox_spatial(I1, I2, **kw)
    gpu64.set_option("sor_exact", mode)
    gpu64.set_option("sor_window", window)
    gpu64.set_option("sor_rows", rows)          # rows per workgroup: a sweep is cut into row blocks
    try:
        ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=12.0, warps=3)
        it_g = list(gpu64.stats().iterations()[0])
        ub, vb = gpu64.brox_spatial(I1, I2, **kw)
        it_b = gpu64.stats().iterations()
    finally:
        gpu64.set_option("sor_exact", 1)
        gpu64.set_option("sor_window", 0)
        gpu64.set_option("sor_rows", 0)
    assert it_g == it_o and np.array_equal(it_b, it_r)
    assert np.abs(ug - uo).max() < 1e-12 and np.abs(vg - vo).max() < 1e-12
    assert np.abs(ub - ur).max() < 1e-11 and np.abs(vb - vr).max() < 1e-11


@pytest.mark.parametrize("nx,ny", [(24, 1100), (20, 1250), (1100, 24), (3, 3), (3, 40), (40, 3)])
def test_exact_windowed_extreme_shapes(gpu64, orc, synth, nx, ny):
    """more plane items than threads of a workgroup (ny + 3 > 1024), and the smallest images with an interior"""
    I1, I2 = synth.pair("P1", nx, ny, 2)
    z = np.zeros((ny, nx))
    uo, vo, it_o = orc.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=2, maxiter=40)
    ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=2, maxiter=40)
    assert list(gpu64.stats().iterations()[0]) == it_o
    assert np.abs(ug - uo).max() < 1e-12 and np.abs(vg - vo).max() < 1e-12
    if min(nx, ny) < 5:
        return          # Brox presmooths (radius 5 > image: "sigma too large" in the reference, operators.cpp:520)
    kw = dict(alpha=30.0, gamma=5.0, nscales=1, nu=0.5, TOL=1e-4, inner=1, outer=2)
    ur, vr, it_r = orc.brox_spatial(I1, I2, **kw)
    ub, vb = gpu64.brox_spatial(I1, I2, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert np.abs(ub - ur).max() < 1e-11 and np.abs(vb - vr).max() < 1e-11


# ---- Brox temporal ----------------------------------------------------------------------------
@pytest.mark.parametrize("nx,ny,frames,kw", [
    (48, 40, 3, dict(nscales=2, outer=3)),                           # nz = 2: no interior frame
    (64, 48, 5, dict(nscales=2, outer=3, inner=2)),
    (33, 47, 4, dict(nscales=1, outer=2, alpha=30.0, gamma=0.0)),    # runs into the 300-sweep limit
    (150, 140, 4, dict(nscales=3, outer=4, nu=0.5)),                 # several row blocks per frame (R = 64)
])
def test_brox_temporal_exact(gpu64, orc, synth, nx, ny, frames, kw):
    I = synth.sequence(nx, ny, frames)
    ur, vr, it_r = orc.brox_temporal(I, **kw)
    ug, vg = gpu64.brox_temporal(I, **kw)
    assert np.array_equal(gpu64.stats().iterations(), it_r)
    assert np.abs(ug - ur).max() < 1e-11 and np.abs(vg - vr).max() < 1e-11


@pytest.mark.parametrize("window,rows", [(1, 2), (5, 3), (32, 0), (8, 1000)])
def test_brox_temporal_any_window(gpu64, orc, synth, window, rows):
    I = synth.sequence(40, 33, 5, 1)
    kw = dict(nscales=1, outer=2, inner=1)
    ur, vr, it_r = orc.brox_temporal(I, **kw)
    gpu64.set_option("sor_window", window)
    gpu64.set_option("sor_rows", rows)
    try:
        ug, vg = gpu64.brox_temporal(I, **kw)
        it_g = gpu64.stats().iterations()
    finally:
        gpu64.set_option("sor_window", 0)
        gpu64.set_option("sor_rows", 0)
    assert np.array_equal(it_g, it_r)
    assert np.abs(ug - ur).max() < 1e-11 and np.abs(vg - vr).max() < 1e-11


def test_brox_temporal_golden_and_errors(gpu64, ofx_mod, synth):
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(here, "cases.json")))
    for name in ("broxt_seq4_64x48", "broxt_seq3_48x40"):
        c, g = cases[name], np.load(os.path.join(here, name + ".npz"))
        u, v = gpu64.brox_temporal(synth.sequence(c["nx"], c["ny"], c["pair"]), **c["params"])
        assert list(gpu64.stats().iterations()[::-1].ravel()) == list(g["iters"])
        assert np.abs(u - g["u"]).max() < 1e-11 and np.abs(v - g["v"]).max() < 1e-11
    with pytest.raises(ofx_mod.OfxError):
        gpu64.brox_temporal(synth.sequence(32, 24, 2))               # "The method needs more than two frames"


# ---- classic Horn-Schunck (Jacobi) ----------------------------------------------------------------------------
@pytest.mark.parametrize("nx,ny,niter", [(64, 48, 50), (135, 68, 200), (33, 21, 7), (5, 4, 3), (2, 2, 1), (320, 240, 0)])
def test_hs_classic_is_bit_exact(gpu64, orc, synth, nx, ny, niter):
    a, b = synth.pair("P1", max(nx, 8), max(ny, 8))
    a, b = np.ascontiguousarray(a[:ny, :nx]), np.ascontiguousarray(b[:ny, :nx])
    uo, vo = orc.hs_classic(a, b, niter, 15.0)
    ug, vg = gpu64.hs_classic(a, b, niter, 15.0)
    assert np.array_equal(ug, uo) and np.array_equal(vg, vo)


def test_hs_classic_f32_and_cli(gpu32, orc, synth, tmp_path):
    import os
    import subprocess
    nx, ny = 96, 64
    a, b = synth.pair("P1", nx, ny)
    uo, vo = orc.hs_classic(a, b, 100, 20.0)
    ug, vg = gpu32.hs_classic(a, b, 100, 20.0)
    assert aepe(ug, vg, uo, vo) < 1e-4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "optical-flow-1_amd", "bin", "horn_schunck_classic")
    for name, img in (("a.pgm", a), ("b.pgm", b)):
        with open(tmp_path / name, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (nx, ny))
            f.write(img.astype(np.uint8).tobytes())
    r = subprocess.run([exe, "100", "20", str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(tmp_path / "f.flo")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "f.flo", "rb").read()
    got = np.frombuffer(raw[12:], dtype=np.float32).reshape(ny, nx, 2)
    assert np.array_equal(got, np.stack([uo, vo], axis=-1).astype(np.float32))       # byte-identical payload
    r = subprocess.run([exe, "100", "20"], capture_output=True, text=True)
    assert "usage:" in r.stderr and r.returncode == len("usage:\n\t%s niter alpha a b f\n" % exe) % 256


# ---- lockstep groups of the SOR solvers (ofx_hs_group_dev / ofx_brox_group_dev / *_batch_dev) ---------------------------
def _group_inputs(synth, G, nx, ny):
    import torch
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    return pairs, d0, d1, flo


@pytest.mark.parametrize("G", [1, 2, 5, 16])
def test_hs_lockstep_group_equals_pairs_solved_alone(gpu64, orc, synth, G):
    """G different pairs through the same windowed launches (blockIdx.z = pair).  Each pair keeps its own error slots,
    snapshots and stopping test: sweep tables equal the oracle's (reference order) pair by pair, and the .flo payloads
    are bit-identical to the flows ofx_hs_pyramidal computes for the pair alone."""
    nx, ny = 150, 97
    kw = dict(alpha=15.0, nscales=3, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
    pairs, d0, d1, flo = _group_inputs(synth, G, nx, ny)
    st = gpu64.hs_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)],
                            nx, ny, **kw)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    seen = set()
    for k in range(G):
        ua, va = gpu64.hs_pyramidal(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), gpu64.stats().iterations()), k
        assert np.array_equal(got[k], np.stack([ua, va], axis=-1).astype(np.float32)), k
        uo, vo, it_o = orc.hs_pyramidal(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), it_o), k
        assert np.abs(ua - uo).max() < 1e-12 and np.abs(va - vo).max() < 1e-12
        seen.add(tuple(int(x) for x in np.asarray(it_o).ravel()))
    if G >= 5:
        assert len(seen) > 1        # the pairs really stop at different sweeps


@pytest.mark.parametrize("G", [1, 3, 16])
def test_brox_lockstep_group_equals_pairs_solved_alone(gpu64, orc, synth, G):
    nx, ny = 140, 90
    kw = dict(alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=2, outer=4)
    pairs, d0, d1, flo = _group_inputs(synth, G, nx, ny)
    st = gpu64.brox_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)],
                              nx, ny, **kw)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    for k in range(G):
        ua, va = gpu64.brox_spatial(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), gpu64.stats().iterations()), k
        assert np.array_equal(got[k], np.stack([ua, va], axis=-1).astype(np.float32)), k
        uo, vo, it_o = orc.brox_spatial(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), it_o), k
        assert np.abs(ua - uo).max() < 1e-11 and np.abs(va - vo).max() < 1e-11


def test_sor_groups_with_small_batches_and_windows(gpu64, synth):
    """pairs that stop in different batches of a solve (small snapshot capacity) and odd window / row-block sizes"""
    nx, ny, G = 96, 70, 4
    kw = dict(alpha=10.0, nscales=2, zfactor=0.5, warps=3, TOL=1e-5, maxiter=90)
    pairs, d0, d1, flo = _group_inputs(synth, G, nx, ny)
    want = [gpu64.hs_pyramidal(p[0], p[1], **kw) + (gpu64.stats().iterations().copy(),) for p in pairs]
    for batch, window, rows, spw in ((5, 3, 16, 1), (9, 8, 64, 2), (300, 5, 7, 4), (7, 8, 125, 2), (64, 4, 200, 4)):
        for name, val in (("sor_batch", batch), ("sor_window", window), ("sor_rows", rows), ("sor_spw", spw)):
            gpu64.set_option(name, val)
        try:
            st = gpu64.hs_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                    [flo[k].data_ptr() for k in range(G)], nx, ny, **kw)
            gpu64.synchronize()
        finally:
            for name in ("sor_batch", "sor_window", "sor_rows", "sor_spw"):
                gpu64.set_option(name, 0)
        got = flo.cpu().numpy()
        for k in range(G):
            assert np.array_equal(st[k].iterations(), want[k][2]), (batch, k)
            assert np.array_equal(got[k], np.stack(want[k][:2], axis=-1).astype(np.float32)), (batch, k)
    # the same for Brox (several sweeps per workgroup, odd geometries)
    bk = dict(alpha=50.0, gamma=10.0, nscales=2, nu=0.5, TOL=1e-4, inner=1, outer=3)
    wantb = [gpu64.brox_spatial(p[0], p[1], **bk) + (gpu64.stats().iterations().copy(),) for p in pairs]
    for batch, window, rows, spw in ((6, 3, 16, 2), (300, 4, 125, 4), (9, 8, 61, 1)):
        for name, val in (("sor_batch", batch), ("sor_window", window), ("sor_rows", rows), ("sor_spw", spw)):
            gpu64.set_option(name, val)
        try:
            st = gpu64.brox_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1],
                                      [flo[k].data_ptr() for k in range(G)], nx, ny, **bk)
            gpu64.synchronize()
        finally:
            for name in ("sor_batch", "sor_window", "sor_rows", "sor_spw"):
                gpu64.set_option(name, 0)
        got = flo.cpu().numpy()
        for k in range(G):
            assert np.array_equal(st[k].iterations(), wantb[k][2]), (batch, k)
            assert np.array_equal(got[k], np.stack(wantb[k][:2], axis=-1).astype(np.float32)), (batch, k)


def test_sor_batch_entry_points(ofx_mod, synth):
    """ofx_hs_batch_dev / ofx_brox_batch_dev: 7 pairs on 2 contexts (groups of 4 + 3), work records per pair"""
    nx, ny, n = 120, 80, 7
    pairs, d0, d1, flo = _group_inputs(synth, n, nx, ny)
    ctxs = [ofx_mod.Ofx(0, ofx_mod.F64) for _ in range(2)]
    solo = ofx_mod.Ofx(0, ofx_mod.F64)
    ptr = lambda ts: [t.data_ptr() for t in ts]
    hk = dict(alpha=20.0, nscales=3, zfactor=0.5, warps=3, TOL=1e-4, maxiter=100)
    work = ofx_mod.hs_batch_dev(ctxs, ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(n)], nx, ny, **hk)
    got = flo.cpu().numpy().copy()
    for k in range(n):
        u, v = solo.hs_pyramidal(pairs[k][0], pairs[k][1], **hk)
        assert np.array_equal(got[k], np.stack([u, v], axis=-1).astype(np.float32)), k
        assert work[k] == solo.stats().work_pix_iters
    bk = dict(alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=1, outer=3)
    work = ofx_mod.brox_batch_dev(ctxs, ptr(d0), ptr(d1), [flo[k].data_ptr() for k in range(n)], nx, ny, **bk)
    got = flo.cpu().numpy().copy()
    for k in range(n):
        u, v = solo.brox_spatial(pairs[k][0], pairs[k][1], **bk)
        assert np.array_equal(got[k], np.stack([u, v], axis=-1).astype(np.float32)), k
        assert work[k] == solo.stats().work_pix_iters
    for c in ctxs + [solo]:
        c.close()


def test_sor_groups_need_the_exact_mode_or_the_tile_sweeps(ofx_mod, gpu64, synth):
    """the one-launch-per-colour kernels (sor_exact = 0, sor_fuse = -1) serve single pairs only; groups in the tolerance
    mode's tile sweeps: tests/test_gpu_sor_tile.py"""
    pairs, d0, d1, flo = _group_inputs(synth, 2, 64, 48)
    gpu64.set_option("sor_exact", 0)
    gpu64.set_option("sor_fuse", -1)
    try:
        with pytest.raises(ofx_mod.OfxError) as e:
            gpu64.hs_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(2)],
                               64, 48, nscales=2)
        assert e.value.status == 1
    finally:
        gpu64.set_option("sor_exact", 1)
        gpu64.set_option("sor_fuse", 0)
    with pytest.raises(ofx_mod.OfxError):
        gpu64.brox_group_dev([], [], [], 64, 48)


# ---- robust_expo_methods, one channel ------------------------------------------------------------------------
@pytest.mark.parametrize("pair,nx,ny,ns,kw", [
    ("P1", 64, 48, 2, dict(method=1, alpha=50.0, gamma=10.0, lam=0.1, outer=4)),
    ("P0", 96, 64, 3, dict(method=2, alpha=18.7, gamma=5.0, lam=0.05, outer=3, inner=2)),
    ("P1", 80, 60, 2, dict(method=3, alpha=30.0, gamma=10.0, lam=1.0, outer=3)),
    ("P1", 33, 21, 1, dict(method=1, alpha=7.9, gamma=0.0, lam=0.3, outer=5)),
    ("P1", 320, 240, 4, dict(method=1, alpha=50.0, gamma=10.0, lam=0.1, outer=5))])
def test_robust_expo_exact(gpu64, orc, synth, pair, nx, ny, ns, kw):
    """ofx_robust_expo against the oracle (== the compiled reference, tests/test_oracle_vs_ref.py): the three decreasing
    functions, alpha truncated to an int, the Dirichlet presmoothing with sigma = 1, the right-left-down-up sums: sweep
    tables equal, flows to < 1e-11 (the order of the stopping sum is the only difference)"""
    I1, I2 = synth.pair(pair, nx, ny)
    uo, vo, it = orc.robust_expo(I1, I2, nscales=ns, **kw)
    ug, vg = gpu64.robust_expo(I1, I2, nscales=ns, **kw)
    got = gpu64.stats().iterations()[:ns, :it.shape[1]]
    assert np.array_equal(got, it)
    assert np.abs(ug - uo).max() < 1e-11 and np.abs(vg - vo).max() < 1e-11


def test_robust_expo_golden_and_errors(gpu64, ofx_mod, synth):
    import json, os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(root, "cases.json")))
    for name in sorted(c for c in cases if cases[c]["kind"] == "rexpo"):
        c, g = cases[name], np.load(os.path.join(root, name + ".npz"))
        I1, I2 = synth.pair(c["pair"], c["nx"], c["ny"])
        u, v = gpu64.robust_expo(I1, I2, **c["params"])
        ns = c["params"]["nscales"]
        assert list(gpu64.stats().iterations()[:ns, :len(g["iters"]) // ns][::-1].ravel()) == list(g["iters"]), name
        assert np.abs(u - g["u"]).max() < 1e-11 and np.abs(v - g["v"]).max() < 1e-11, name
    z = np.zeros((32, 32))
    for bad in (dict(nz=3), dict(method=0), dict(method=4), dict(inner=-1)):
        with pytest.raises(ofx_mod.OfxError):
            gpu64.robust_expo(z, z, **bad)
