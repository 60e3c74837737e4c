"""CPU-side logic of the product: image I/O of the front-ends, argument handling that needs no GPU,
the hypot restatement the kernels rely on, and the synthetic-input generator."""
import ctypes as C
import math
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optical-flow-1_amd")


@pytest.fixture(scope="module")
def io():
    so = os.path.join(PKG, "libofxio.so")
    if not os.path.exists(so) or not os.path.exists(os.path.join(PKG, "bin", "tvl1flow")):
        if not os.path.exists(os.path.join(PKG, "libofx.so")):
            subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], check=True, capture_output=True)
        subprocess.run(["make", "-C", os.path.join(PKG, "cli")], check=True, capture_output=True)
    L = C.CDLL(so)
    L.ofx_read_image_double.restype = C.POINTER(C.c_double)
    L.ofx_read_image_double.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ofx_read_flo.restype = C.POINTER(C.c_float)
    L.ofx_read_flo.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ofx_write_flo.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
    return L


def read_image(io, path):
    w, h = C.c_int(), C.c_int()
    p = io.ofx_read_image_double(str(path).encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()


def test_pgm_binary_ascii_and_comments(io, tmp_path):
    img = (np.arange(35).reshape(5, 7) * 7 % 256).astype(np.uint8)
    (tmp_path / "a.pgm").write_bytes(b"P5\n# a comment\n7 5\n255\n" + img.tobytes())
    assert np.array_equal(read_image(io, tmp_path / "a.pgm"), img.astype(np.float64))
    (tmp_path / "b.pgm").write_text("P2\n7 5 # w h\n255\n" + " ".join(str(int(x)) for x in img.ravel()) + "\n")
    assert np.array_equal(read_image(io, tmp_path / "b.pgm"), img.astype(np.float64))
    # maxval is ignored except for the sample width: 16-bit samples are big-endian (iio.cpp:1744-1755)
    big = (np.arange(35).reshape(5, 7) * 997 % 65536).astype(">u2")
    (tmp_path / "c.pgm").write_bytes(b"P5\n7 5\n65535\n" + big.tobytes())
    assert np.array_equal(read_image(io, tmp_path / "c.pgm"), big.astype(np.float64))
    assert read_image(io, tmp_path / "missing.pgm") is None


def test_ppm_is_collapsed_to_gray_in_float(io, tmp_path):
    rgb = np.random.default_rng(0).integers(0, 256, (4, 6, 3)).astype(np.uint8)
    (tmp_path / "a.ppm").write_bytes(b"P6\n6 4\n255\n" + rgb.tobytes())
    f = rgb.astype(np.float32).astype(np.float64)
    want = (.299 * f[..., 0] + .587 * f[..., 1] + .114 * f[..., 2]).astype(np.float32).astype(np.float64)
    assert np.array_equal(read_image(io, tmp_path / "a.ppm"), want)          # iio.cpp:1110-1118


def test_pfm_is_read_without_flip_or_swap(io, tmp_path):
    data = np.random.default_rng(1).standard_normal((3, 5)).astype(np.float32)
    (tmp_path / "a.pfm").write_bytes(b"Pf\n5 3\n-1.0\n" + data.tobytes())
    assert np.array_equal(read_image(io, tmp_path / "a.pfm"), data.astype(np.float64))   # iio.cpp:2194-2229


def test_flo_layout(io, tmp_path):
    uv = np.random.default_rng(2).standard_normal((4, 6, 2)).astype(np.float32)
    path = tmp_path / "x.flo"
    assert io.ofx_write_flo(str(path).encode(), uv.ctypes.data_as(C.POINTER(C.c_float)), 6, 4) == 0
    raw = path.read_bytes()
    assert raw[:4] == b"PIEH" and struct.unpack("<f", raw[:4])[0] == 202021.25       # iio.cpp:2753-2777
    assert struct.unpack("<II", raw[4:12]) == (6, 4)
    assert np.array_equal(np.frombuffer(raw[12:], dtype=np.float32).reshape(4, 6, 2), uv)
    w, h = C.c_int(), C.c_int()
    p = io.ofx_read_flo(str(path).encode(), C.byref(w), C.byref(h))
    assert (w.value, h.value) == (6, 4)
    assert np.array_equal(np.ctypeslib.as_array(p, shape=(4, 6, 2)), uv)


def test_front_ends_usage_and_exit_codes(io, tmp_path):
    b = os.path.join(PKG, "bin")
    r = subprocess.run([os.path.join(b, "tvl1flow")], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "nproc tau lambda theta nscales zfactor nwarps epsilon verbose" in r.stderr
    r = subprocess.run([os.path.join(b, "horn_schunck_pyramidal"), "only_one"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr
    r = subprocess.run([os.path.join(b, "brox_spatial")], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage:" in r.stdout          # brox_spatial_main.cpp always returns 0 (:195)
    r = subprocess.run([os.path.join(b, "tvl1flow"), str(tmp_path / "no.pgm"), str(tmp_path / "no2.pgm")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and 'could not read image from file' in r.stderr
    (tmp_path / "a.pgm").write_bytes(b"P5\n4 4\n255\n" + bytes(16))
    (tmp_path / "b.pgm").write_bytes(b"P5\n5 4\n255\n" + bytes(20))
    r = subprocess.run([os.path.join(b, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "size mismatch 4x4 != 5x4" in r.stderr


def hypot_restated(x, y):
    """The operation sequence of ofx_device.h:hypot_ref (glibc >= 2.35 non-FMA algorithm)."""
    x, y = abs(x), abs(y)
    ax, ay = (y, x) if x < y else (x, y)
    SCALE, LARGE, TINY, EPS = 2.0 ** -600, 2.0 ** 511, 2.0 ** -459, 2.0 ** -54

    def kernel(ax, ay):
        h = math.sqrt(ax * ax + ay * ay)
        if h <= 2.0 * ay:
            d = h - ay
            t1 = ax * (2.0 * d - ax)
            t2 = (d - 2.0 * (ax - ay)) * d
        else:
            d = h - ax
            t1 = 2.0 * d * (ax - 2.0 * ay)
            t2 = (4.0 * d - ay) * ay + d * d
        return h - (t1 + t2) / (2.0 * h)

    if ax > LARGE:
        return ax + ay if ay <= ax * EPS else kernel(ax * SCALE, ay * SCALE) / SCALE
    if ay < TINY:
        return ax + ay if ax >= ay / EPS else kernel(ax / SCALE, ay / SCALE) * SCALE
    if ax >= ay / EPS:
        return ax + ay
    return kernel(ax, ay)


def test_hypot_restatement_matches_libm():
    rng = np.random.default_rng(3)
    n = 200000
    xs = np.ldexp(rng.uniform(-1, 1, n), rng.integers(-60, 20, n))
    ys = np.ldexp(rng.uniform(-1, 1, n), rng.integers(-60, 20, n))
    ys[::7] = xs[::7] * rng.uniform(0.5, 2.0, len(xs[::7]))
    ys[::1000] = 0.0
    libm = C.CDLL("libm.so.6")          # glibc's hypot -- what the reference calls (CPython's math.hypot is its own code)
    libm.hypot.restype = C.c_double
    libm.hypot.argtypes = [C.c_double, C.c_double]
    bad = sum(1 for x, y in zip(xs, ys) if hypot_restated(float(x), float(y)) != libm.hypot(float(x), float(y)))
    assert bad == 0


def test_synthetic_pairs_are_stable(synth):
    I0, I1 = synth.pair("P0", 64, 48)
    assert I0.shape == (48, 64) and I0.min() >= 0 and I0.max() <= 255 and np.array_equal(I0, np.floor(I0))
    assert (int(I0.sum()), int(I1.sum())) == (int(synth.pair_p0(64, 48)[0].sum()), int(synth.pair_p0(64, 48)[1].sum()))
    a0, _ = synth.pair("P1", 64, 48, 0)
    a1, _ = synth.pair("P1", 64, 48, 1)
    assert not np.array_equal(a0, a1)


def test_windowed_sor_schedule_is_valid():
    """the (sweep, row block, step) schedule of the windowed exact SOR kernels never reads a value of another
    workgroup from the same launch, and never has one overwritten in it (tools/check_sor_schedule.py)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_sor_schedule", os.path.join(ROOT, "tools", "check_sor_schedule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for solver in ("hs", "brox"):
        for nx, ny, K, B in [(23, 30, 8, 3), (9, 12, 4, 5), (12, 40, 1, 4), (7, 9, 8, 1), (16, 9, 8, 2)]:
            assert mod.check(nx, ny, K, B, solver, 2) == 0


def test_lds_windows_of_the_sor_kernels_cover_every_access():
    """k_hs_window_lds / k_brox_window_lds copy a launch's working set into LDS when the launch starts.  The windows they
    stage -- HsWinLds: unknowns on hyperplanes q0 - 7 .. q0 + K + 2, coefficients q0 - 7 .. q0 + K - 1; BroxWinLds: q0 - 4 ..
    q0 + K and q0 - 4 .. q0 + K - 1; rows b R - 2 .. b R + R -- must contain everything the K steps touch, border pixels
    included (tools/check_sor_schedule.py cover() enumerates the accesses; also for K = 16 / 24, the new defaults)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_sor_schedule", os.path.join(ROOT, "tools", "check_sor_schedule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for solver, back, ahead in (("hs", 7, 2), ("brox", 4, 0)):
        for nx, ny, R in [(23, 30, 8), (9, 12, 3), (40, 17, 5), (5, 5, 2), (3, 3, 2), (12, 40, 64)]:
            for K in (4, 8, 16, 24):
                hmin, hmax, rmin, rmax, cmin, cmax = mod.cover(nx, ny, K, R, solver)
                assert hmin >= -back and hmax <= K + ahead, (solver, nx, ny, R, K, hmin, hmax)
                assert rmin >= -2 and rmax <= R, (solver, nx, ny, R, K, rmin, rmax)
                assert cmin >= -back and cmax <= K - 1, (solver, nx, ny, R, K, cmin, cmax)
    for solver in ("hs", "brox"):                          # the schedule itself with the larger windows
        for nx, ny, K, B in [(23, 30, 16, 3), (16, 9, 24, 2), (9, 12, 16, 1)]:
            assert mod.check(nx, ny, K, B, solver, 2) == 0
    # sor_unit_idle: a (sweep, row block) unit leaves a launch at once when the launch's steps lie outside the steps at which
    # the block has pixels -- no pixel's step may lie outside the range assumed for the block that executes it
    for solver in ("hs", "brox"):
        for nx, ny in [(23, 52), (40, 31), (17, 9), (9, 33), (5, 5), (3, 3), (2, 7), (7, 2), (4, 3), (100, 20)]:
            for R in (2, 3, 5, 7, 16, 64):
                assert mod.unit_range_violations(nx, ny, R, solver) == 0, (solver, nx, ny, R)


def test_rof_iteration_pipeline_schedule():
    """k_rof_window keeps all iterations of a Scalar_ROF_BoxCellCentered call in flight: sweep s follows ROF_LAGI = 120 positions
    behind sweep s - 1, the alfa stage of s runs ROF_D = 58 positions ahead of it (ofx_occ.hip).  tools/check_rof_pipeline.py
    enumerates, per cell, what each stage takes from another workgroup: stored by an earlier launch, overwritten by a later one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_rof_pipeline", os.path.join(ROOT, "tools", "check_rof_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = open(os.path.join(ROOT, "optical-flow-1_amd", "csrc", "ofx_occ.hip")).read()
    assert "#define ROF_K 24" in src and "#define ROF_LAG (ROF_K + 8)" in src and "#define ROF_LAGI 120" in src
    assert "#define ROF_D (ROF_K + ROF_LAG + 2)" in src and "#define ROF_NT 128" in src and "#define ROF_R (ROF_NT - 3)" in src
    assert "#define ROF_D_1 (ROF_K + 2)" in src and "#define ROF_LAGI_1 56" in src
    for nx, ny, R in [(40, 30, 125), (17, 260, 125), (9, 9, 125), (2, 2, 125), (33, 20, 7), (5, 40, 3)]:
        assert mod.violations(nx, ny, R, 24, 32, 120, 58) == 0, (nx, ny, R)
    for nx, ny in [(40, 30), (160, 120), (9, 9), (2, 2), (3, 125), (125, 3)]:            # one row block: the shorter lags
        assert mod.violations(nx, ny, 125, 24, 32, 56, 26) == 0, (nx, ny)
    assert mod.violations(17, 260, 125, 24, 32, 56, 26) > 0                              # ... which several blocks do not allow
    # the 10-step window of lockstep groups (RofGeo<10>): blocks 18 apart, D = 30 / 12, LAGI = 64 / 28
    assert "#define ROF_KG 10" in src and "static constexpr int D = K + LAG + 2, LAGI = D + K + LAG + 6;" in src
    assert "static constexpr int D1 = K + 2, LAGI1 = D1 + K + 6;" in src
    for nx, ny, R in [(40, 30, 125), (17, 260, 125), (9, 9, 125), (33, 20, 7), (5, 40, 3)]:
        assert mod.violations(nx, ny, R, 10, 18, 64, 30) == 0, (nx, ny, R)
    for nx, ny in [(40, 30), (160, 120), (2, 2), (3, 125)]:
        assert mod.violations(nx, ny, 125, 10, 18, 28, 12) == 0, (nx, ny)
    assert mod.violations(33, 20, 7, 10, 18, 58, 30) > 0
    assert mod.violations(33, 20, 7, 24, 32, 112, 58) > 0 and mod.violations(33, 20, 7, 24, 32, 120, 50) > 0    # the checker does bite


def test_flo_reader_on_the_reference_own_flow_file(io):
    """tests/golden/ipol_tvl1flow_3_uv.flo is the one data file the reference ships (3rdparty/tvl1flow_3/uv.flo,
    a 256x256 flow written by the IPOL original's iio; its input images were removed upstream, so it pins the
    .flo layout our reader / writer use, not a solver)."""
    path = os.path.join(ROOT, "tests", "golden", "ipol_tvl1flow_3_uv.flo")
    raw = open(path, "rb").read()
    assert raw[:4] == b"PIEH" and struct.unpack("<II", raw[4:12]) == (256, 256) and len(raw) == 12 + 256 * 256 * 8
    want = np.frombuffer(raw[12:], dtype=np.float32).reshape(256, 256, 2)
    w, h = C.c_int(), C.c_int()
    p = io.ofx_read_flo(path.encode(), C.byref(w), C.byref(h))
    assert (w.value, h.value) == (256, 256)
    got = np.ctypeslib.as_array(p, shape=(256, 256, 2))
    assert np.array_equal(got, want) and np.isfinite(got).all()


def _png(path, w, h, depth, color_type, rows, palette=None):
    """minimal PNG writer (filter 0 rows): color_type 0 gray, 2 RGB, 3 palette, 4 gray+alpha, 6 RGBA"""
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    out += chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


def test_png_input_follows_iio_semantics(io, tmp_path):
    """PNG through the run-time-bound libpng: 8-bit gray as is; RGB -> (uint8)(.299R+.587G+.114B) in double arithmetic
    (iio.cpp:1100-1108); palettes expanded to RGB first; 4-bit gray scaled to 8 bit (PNG_TRANSFORM_EXPAND); 16-bit gray
    as host-order uint16; alpha / 16-bit colour are rejected like in the reference (iio.cpp:3595-3597,1119-1121).
    Parity unpinned against iio itself (it cannot be built here: no libpng headers) -- restated from its source."""
    import ctypes.util
    if not (ctypes.util.find_library("png16") or os.path.exists("/lib/x86_64-linux-gnu/libpng16.so.16")):
        pytest.skip("libpng16 not present on this machine")
    rng = np.random.default_rng(5)
    w, h = 7, 5
    g8 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    _png(tmp_path / "g8.png", w, h, 8, 0, g8)
    assert np.array_equal(read_image(io, tmp_path / "g8.png"), g8.astype(np.float64))
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    _png(tmp_path / "rgb.png", w, h, 8, 2, rgb.reshape(h, w * 3))
    want = (.299 * rgb[..., 0].astype(np.float64) + .587 * rgb[..., 1] + .114 * rgb[..., 2]).astype(np.uint8)
    assert np.array_equal(read_image(io, tmp_path / "rgb.png"), want.astype(np.float64))
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    _png(tmp_path / "pal.png", w, h, 8, 3, idx, palette=pal.reshape(-1))
    p = pal[idx]
    want = (.299 * p[..., 0].astype(np.float64) + .587 * p[..., 1] + .114 * p[..., 2]).astype(np.uint8)
    assert np.array_equal(read_image(io, tmp_path / "pal.png"), want.astype(np.float64))
    g4 = rng.integers(0, 16, (h, 8), dtype=np.uint8)                       # 8 pixels per row, two per byte
    _png(tmp_path / "g4.png", 8, h, 4, 0, (g4[:, 0::2] << 4) | g4[:, 1::2])
    assert np.array_equal(read_image(io, tmp_path / "g4.png"), (g4 * 17).astype(np.float64))
    g16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    _png(tmp_path / "g16.png", w, h, 16, 0, g16.astype(">u2").view(np.uint8).reshape(h, w * 2))
    assert np.array_equal(read_image(io, tmp_path / "g16.png"), g16.astype(np.float64))
    rgba = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    _png(tmp_path / "rgba.png", w, h, 8, 6, rgba)
    wv, hv = C.c_int(), C.c_int()
    assert not io.ofx_read_image_double(str(tmp_path / "rgba.png").encode(), C.byref(wv), C.byref(hv))
    (tmp_path / "broken.png").write_bytes(open(tmp_path / "g8.png", "rb").read()[:40])
    assert not io.ofx_read_image_double(str(tmp_path / "broken.png").encode(), C.byref(wv), C.byref(hv))


def test_png_decoding_agrees_with_an_independent_decoder(io, tmp_path):
    """SURVEY 8(f)2, still "parity UNPINNED" against iio (which cannot be built here and ships no image fixtures): what CAN be
    pinned is the decoding itself.  Pillow writes the files (real encoder: filters, compression, interlacing) and decodes them
    again; ofx_read_image_double's samples are compared with Pillow's -- 8- and 16-bit gray as is, RGB and palette images through
    the iio collapse (uint8)(.299 R + .587 G + .114 B) applied to Pillow's RGB samples."""
    import ctypes.util
    PIL = pytest.importorskip("PIL.Image")
    if not (ctypes.util.find_library("png16") or os.path.exists("/lib/x86_64-linux-gnu/libpng16.so.16")):
        pytest.skip("libpng16 not present on this machine")
    rng = np.random.default_rng(11)
    w, h = 37, 23
    collapse = lambda rgb: (.299 * rgb[..., 0].astype(np.float64) + .587 * rgb[..., 1] + .114 * rgb[..., 2]).astype(np.uint8)
    g8 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    PIL.fromarray(g8).save(tmp_path / "g8.png", optimize=True)
    assert np.array_equal(read_image(io, tmp_path / "g8.png"), np.asarray(PIL.open(tmp_path / "g8.png")).astype(np.float64))
    g16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    PIL.fromarray(g16).save(tmp_path / "g16.png")
    assert np.array_equal(read_image(io, tmp_path / "g16.png"), np.asarray(PIL.open(tmp_path / "g16.png")).astype(np.float64))
    assert np.array_equal(read_image(io, tmp_path / "g16.png"), g16.astype(np.float64))
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    PIL.fromarray(rgb).save(tmp_path / "rgb.png", compress_level=9)
    back = np.asarray(PIL.open(tmp_path / "rgb.png").convert("RGB"))
    assert np.array_equal(back, rgb)
    assert np.array_equal(read_image(io, tmp_path / "rgb.png"), collapse(back).astype(np.float64))
    pal = PIL.fromarray(rgb).quantize(colors=16)                     # a palette image with Pillow's own palette
    pal.save(tmp_path / "pal.png")
    back = np.asarray(PIL.open(tmp_path / "pal.png").convert("RGB"))
    assert np.array_equal(read_image(io, tmp_path / "pal.png"), collapse(back).astype(np.float64))


def test_bench_round_split_and_self_launch_command(monkeypatch):
    """bench.py: rounds are a partition with the last round ~1/4 (multiple of the contexts); N > 1 without a launcher
    starts torch.distributed.run as a child BEFORE torch is imported."""
    import bench
    for n in (1, 2, 3, 5, 8, 20, 64):
        for r in (1, 2, 3):
            sl = bench.split_rounds(n, r, 4)
            assert sl[0][0] == 0 and sum(c for _, c in sl) == n and all(c >= 1 for _, c in sl)
            assert all(sl[i][0] + sl[i][1] == sl[i + 1][0] for i in range(len(sl) - 1))
    assert bench.split_rounds(64, 2, 4) == [(0, 48), (48, 16)]
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20"])

    class A:
        gpus = 4

    assert bench.self_launch(A()) == 7
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in c and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "4", "--steps", "20"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gray_byte_image_writer_pgm_and_png(io, tmp_path):
    """iio_save_image_float on a byte-valued one-channel image (the occlusion map of tvl1occflow): P2 up to 10000 pixels, P5
    above, 8-bit gray PNG for .png names (src/iio.cpp:3698-3710,3752-3773,3838-3853); TIFF names and non-byte samples refused"""
    import zlib
    io.ofx_write_gray_bytes.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
    rng = np.random.default_rng(5)

    def write(path, img):
        a = np.ascontiguousarray(img, dtype=np.float32)
        return io.ofx_write_gray_bytes(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), img.shape[1], img.shape[0])
    small = (rng.integers(0, 2, (50, 80)) * 255).astype(np.float32)
    assert write(tmp_path / "s.pgm", small) == 0
    txt = (tmp_path / "s.pgm").read_text()
    assert txt.startswith("P2\n80 50\n255\n") and txt.split("\n")[3:-1] == [str(int(v)) for v in small.ravel()]
    big = rng.integers(0, 256, (101, 100)).astype(np.float32)
    assert write(tmp_path / "b.whatever", big) == 0
    raw = (tmp_path / "b.whatever").read_bytes()
    assert raw == b"P5\n100 101\n255\n" + big.astype(np.uint8).tobytes()
    assert write(tmp_path / "x.tiff", small) == 2 and write(tmp_path / "x.pgm", small + 0.5) == 2
    assert write(tmp_path / "x.pgm", small - 1.0) == 2 and not (tmp_path / "x.pgm").exists()
    r = write(tmp_path / "o.png", big)
    if r != 0:
        pytest.skip("libpng16 not present on this machine")
    data = (tmp_path / "o.png").read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">IIBBBBB", data[16:29]) == (100, 101, 8, 0, 0, 0, 0)
    idat, pos = b"", 8
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        if tag == b"IDAT":
            idat += data[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(101, 101)
    # undo the per-row filters (libpng picks them adaptively); only None(0) and Sub(1) and Up(2) etc. via generic reconstruction
    out = np.zeros((101, 100), dtype=np.uint8)
    for y in range(101):
        ft, line = rows[y, 0], rows[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(100, dtype=np.int32)
        cur = np.zeros(100, dtype=np.int32)
        for x in range(100):
            a = cur[x - 1] if x else 0
            b, c = prev[x], (prev[x - 1] if x else 0)
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[x] = (line[x] + p) & 255
        out[y] = cur
    assert np.array_equal(out, big.astype(np.uint8))
    assert np.array_equal(read_image(io, tmp_path / "o.png"), big.astype(np.float64))       # and through our own reader


def test_staging_route_of_the_cross_gpu_batch():
    """ofx_tvl1_batch_dev over contexts on several GPUs: where a buffer that is not on the context's GPU comes from.  The decision is a
    pure function of (context device, memory device, hipDeviceCanAccessPeer) -- the path itself has never run on more than one GPU."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "optical-flow-1_amd", "libofx.so"))
    f = lib.ofx_staging_route
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 3
    assert f(0, 0, 1) == 0 and f(3, 3, 0) == 0             # on the context's GPU: in place
    assert f(0, -1, 1) == 1 and f(0, -1, 0) == 1           # host memory: one copy
    assert f(0, 1, 1) == 1                                 # another GPU, peer access: one copy over xGMI
    assert f(0, 1, 0) == 2 and f(5, 2, 0) == 2             # no peer access: through a host bounce buffer
