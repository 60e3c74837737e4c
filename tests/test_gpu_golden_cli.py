"""GPU path against the committed golden vectors (outputs of the compiled reference) and through the
drop-in command-line front-ends (PGM in, .flo out)."""
import json
import os
import subprocess

import numpy as np
import pytest
from conftest import require_or_skip, aepe

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "optical-flow-1_amd", "bin")
CASES = json.load(open(os.path.join(G, "cases.json")))


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("tag", ["7x5", "16x16", "135x68"])
def test_operators_vs_reference_vectors(gpu64, tag):
    g = load("operators")
    a, b, img, u, v = (g["in_%s_%s" % (k, tag)] for k in ("a", "b", "img", "u", "v"))
    ny, nx = a.shape
    assert np.array_equal(gpu64.divergence(a, b), g["divergence_" + tag])
    fx, fy = gpu64.forward_gradient(a)
    assert np.array_equal(fx, g["fwd_x_" + tag]) and np.array_equal(fy, g["fwd_y_" + tag])
    cx, cy = gpu64.centered_gradient(a)
    assert np.array_equal(cx, g["cen_x_" + tag]) and np.array_equal(cy, g["cen_y_" + tag])
    for name in ("dxx", "dyy", "dxy"):
        assert np.array_equal(getattr(gpu64, name)(a), g[name + "_" + tag])
    assert np.array_equal(gpu64.bicubic_warp(img, u, v, True), g["warp_bo_" + tag])
    assert np.array_equal(gpu64.bicubic_warp(img, u, v, False), g["warp_nb_" + tag])
    n1, n2 = gpu64.image_normalization_2(img, img * 0.5 + 3)
    assert np.array_equal(n1, g["norm1_" + tag]) and np.array_equal(n2, g["norm2_" + tag])
    if "gauss08_" + tag in g:
        assert np.array_equal(gpu64.gaussian(img, 0.8), g["gauss08_" + tag])
        assert np.array_equal(gpu64.zoom_out(img, 0.5), g["zoomout05_" + tag])
        assert np.array_equal(gpu64.zoom_out(img, 0.7), g["zoomout07_" + tag])
        assert np.array_equal(gpu64.zoom_in(a, 2 * nx - 1, 2 * ny), g["zoomin_" + tag])
    if tag == "7x5":
        pts = g["at_points"]
        assert np.array_equal(gpu64.bicubic_at(img, pts[:, 0], pts[:, 1], False), g["at_nb"])
        assert np.array_equal(gpu64.bicubic_at(img, pts[:, 0], pts[:, 1], True), g["at_bo"])


@pytest.mark.parametrize("case", sorted(c for c in CASES if CASES[c]["kind"] == "tvl1"))
def test_tvl1_vs_reference_vectors(gpu64, synth, case):
    c, g = CASES[case], load(case)
    I0, I1 = synth.pair(c["pair"], c["nx"], c["ny"])
    u, v = gpu64.tvl1_multiscale(I0, I1, **c["params"])
    assert list(gpu64.stats().iterations()[::-1].ravel()) == list(g["iters"])     # reference prints coarse -> fine
    assert aepe(u, v, g["u"], g["v"]) < 1e-4
    assert np.abs(u - g["u"]).max() < 1e-9 and np.abs(v - g["v"]).max() < 1e-9


def test_hs_and_brox_vs_reference_vectors(gpu64, synth):
    c, g = CASES["hs_p1_96x64"], load("hs_p1_96x64")
    I0, I1 = synth.pair(c["pair"], c["nx"], c["ny"])
    u, v = gpu64.hs_pyramidal(I0, I1, **c["params"])
    assert list(gpu64.stats().iterations()[::-1].ravel()) == list(g["iters"])
    assert aepe(u, v, g["u"], g["v"]) < 1e-4 and np.abs(u - g["u"]).max() < 1e-12
    c, g = CASES["brox_p1_96x64"], load("brox_p1_96x64")
    u, v = gpu64.brox_spatial(I0, I1, **c["params"])
    assert list(gpu64.stats().iterations()[::-1].ravel()) == list(g["iters"])
    assert aepe(u, v, g["u"], g["v"]) < 1e-4 and np.abs(u - g["u"]).max() < 1e-11


@pytest.mark.parametrize("case", sorted(c for c in CASES if CASES[c]["kind"] == "occ"))
def test_tvl1occ_vs_reference_vectors(gpu64, synth, case):
    """TV-L1 with occlusions against the compiled reference's committed outputs (zero-filled heap): flow, occlusion map and the
    outer iterations the reference printed, bit for bit"""
    c, g = CASES[case], load(case)
    seq = synth.sequence(c["nx"], c["ny"], 3, c["pair"])
    u, v, chi = gpu64.tvl1occ_multiscale(seq[0], seq[1], seq[2], **c["params"])
    assert np.array_equal(u, g["u"]) and np.array_equal(v, g["v"]) and np.array_equal(chi, g["chi"])
    st, P = gpu64.stats(), c["params"]
    got = [st.iters[s][w] for s in range(P["nscales"] - 1, -1, -1) for w in range(P["warps"])]      # printed coarse to fine
    assert got == list(g["iters"])


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.astype(np.uint8).tobytes())


def read_flo(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"PIEH"
    w, h = np.frombuffer(raw[4:12], dtype=np.uint32)
    return np.frombuffer(raw[12:], dtype=np.float32).reshape(int(h), int(w), 2)


def test_tvl1flow_cli_is_drop_in(orc, synth, tmp_path):
    nx, ny = 160, 120
    I0, I1 = synth.pair("P1", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    out = tmp_path / "o.flo"
    #            I0 I1 out nproc tau lambda theta nscales zfactor nwarps epsilon verbose
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(out),
                        "0", "0.25", "0.15", "0.3", "100", "0.5", "5", "0.01", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # auto-nscales: N = 1 + log(hypot(160,120)/16)/log(2) = 4.64 -> 4 (src/tvl1flow_main.cpp:185-188)
    assert "nscales=4 " in r.stderr and "Scale 3: 20x15" in r.stderr
    uo, vo, it, _ = orc.tvl1_multiscale(I0, I1, nscales=4)
    want = np.stack([uo, vo], axis=-1).astype(np.float32)
    assert np.array_equal(read_flo(out), want)                      # byte-identical payload
    printed = [int(x.split("Iterations: ")[1].split(",")[0]) for x in r.stderr.splitlines() if "Iterations:" in x]
    assert printed == list(it[::-1].ravel())
    # out-of-range arguments silently fall back to the defaults (:102-167)
    r2 = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"),
                         str(tmp_path / "o2.flo"), "-3", "9", "-1", "0", "-5", "1.5", "0", "-1"], capture_output=True)
    assert r2.returncode == 0
    assert np.array_equal(read_flo(tmp_path / "o2.flo"), want)


def test_other_front_ends_run(synth, tmp_path):
    nx, ny = 96, 64
    I0, I1 = synth.pair("P1", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    r = subprocess.run([os.path.join(BIN, "horn_schunck_pyramidal"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"),
                        str(tmp_path / "h.flo"), "0", "20", "3", "0.5", "4", "0.0001", "150", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Scale: 2 24x16" in r.stderr
    g = load("hs_p1_96x64")
    f = read_flo(tmp_path / "h.flo")
    assert np.array_equal(f, np.stack([g["u"], g["v"]], axis=-1).astype(np.float32))     # byte-identical .flo
    r = subprocess.run([os.path.join(BIN, "brox_spatial"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"),
                        str(tmp_path / "x.flo"), "0", "50", "10", "3", "0.5", "0.0001", "1", "4", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Scale: 2" in r.stdout and "Iterations:" in r.stdout
    # Brox clamps nscales with min(nx,ny): N = 1 + log2(64/16) = 3 -> 3 scales
    g = load("brox_p1_96x64")
    f = read_flo(tmp_path / "x.flo")
    assert np.array_equal(f, np.stack([g["u"], g["v"]], axis=-1).astype(np.float32))


def test_sor_front_ends_in_the_tolerance_mode(orc, synth, tmp_path):
    """OFX_SOR_TOLERANCE=1: the SOR front-ends with the re-ordered sweeps of option sor_exact = 0 -- the .flo equals the oracle's
    restatement of those sweep orders bit for bit (and differs from the exact mode's, which is the reference's)"""
    nx, ny = 200, 140
    I0, I1 = synth.pair("P1", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    env = dict(os.environ, OFX_SOR_TOLERANCE="1")
    orc.set_sor_order(1)
    orc.set_sor_wave_levels(1)
    try:
        r = subprocess.run([os.path.join(BIN, "horn_schunck_pyramidal"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"),
                            str(tmp_path / "h.flo"), "0", "20", "3", "0.5", "4", "0.0001", "150", "0"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        uo, vo, _ = orc.hs_pyramidal(I0, I1, alpha=20.0, nscales=3, zfactor=0.5, warps=4, TOL=1e-4, maxiter=150)
        assert np.array_equal(read_flo(tmp_path / "h.flo"), np.stack([uo, vo], axis=-1).astype(np.float32))
        r = subprocess.run([os.path.join(BIN, "brox_spatial"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"),
                            str(tmp_path / "x.flo"), "0", "50", "10", "3", "0.5", "0.0001", "1", "4", "0"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        uo, vo, _ = orc.brox_spatial(I0, I1, alpha=50.0, gamma=10.0, nscales=3, nu=0.5, TOL=1e-4, inner=1, outer=4)
        assert np.array_equal(read_flo(tmp_path / "x.flo"), np.stack([uo, vo], axis=-1).astype(np.float32))
    finally:
        orc.set_sor_order(0)
        orc.set_sor_wave_levels(0)


def test_batch_front_end_writes_the_same_flo_as_tvl1flow(orc, synth, tmp_path):
    """optical-flow-1_amd/batch_run.py on one GPU: 3 pairs from PGM files -> 3 .flo files, byte-identical to
    the single-pair results (and, rehearsing the 2-rank path over gloo on the same GPU, identical again)."""
    import sys
    nx, ny = 128, 96
    lines = []
    want = []
    for k in range(3):
        I0, I1 = synth.pair("P1", nx, ny, k)
        write_pgm(tmp_path / ("a%d.pgm" % k), I0)
        write_pgm(tmp_path / ("b%d.pgm" % k), I1)
        lines.append("%s %s out%d.flo" % (tmp_path / ("a%d.pgm" % k), tmp_path / ("b%d.pgm" % k), k))
        uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=4)      # auto: 1 + log2(160/16) = 4.32 -> 4
        want.append(np.stack([uo, vo], axis=-1).astype(np.float32))
    (tmp_path / "pairs.txt").write_text("\n".join(lines) + "\n")
    script = os.path.join(ROOT, "optical-flow-1_amd", "batch_run.py")
    r = subprocess.run([sys.executable, script, "--list", str(tmp_path / "pairs.txt"), "--out-dir", str(tmp_path / "o1")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in range(3):
        assert np.array_equal(read_flo(tmp_path / "o1" / ("out%d.flo" % k)), want[k])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", script, "--list", str(tmp_path / "pairs.txt"),
                        "--out-dir", str(tmp_path / "o2"), "--backend", "gloo"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in range(3):
        assert np.array_equal(read_flo(tmp_path / "o2" / ("out%d.flo" % k)), want[k])


def test_tvl1flow_cli_reads_png(synth, tmp_path):
    """the same pair as 8-bit gray PNG: identical .flo to PGM input; as an RGB PNG with r = g = b: identical to PGM
    input of the collapsed image (uint8)(.299x+.587x+.114x), which is x - 1 for 65 of the 256 gray levels because
    the double sum lands just below x (iio.cpp:1104-1105)"""
    import struct
    import zlib

    def png(path, img, rgb):
        h, w = img.shape
        rows = np.repeat(img[:, :, None], 3, axis=2).reshape(h, w * 3) if rgb else img

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
        raw = b"".join(b"\x00" + r.astype(np.uint8).tobytes() for r in rows)
        open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if rgb else 0, 0, 0, 0))
                               + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    nx, ny = 96, 64
    I0, I1 = synth.pair("P1", nx, ny)
    args = ["0", "0.25", "0.15", "0.3", "3", "0.5", "5", "0.01", "0"]
    def collapse(img):          # iio.cpp:1104-1105 on r = g = b = img
        x = img.astype(np.float64)
        return (.299 * x + .587 * x + .114 * x).astype(np.uint8)
    flows = {}
    for kind in ("pgm", "gray", "rgb", "pgm_collapsed"):
        ext = "pgm" if kind.startswith("pgm") else "png"
        a, b = tmp_path / ("a_%s.%s" % (kind, ext)), tmp_path / ("b_%s.%s" % (kind, ext))
        if kind == "pgm":
            write_pgm(a, I0)
            write_pgm(b, I1)
        elif kind == "pgm_collapsed":
            write_pgm(a, collapse(I0.astype(np.uint8)))
            write_pgm(b, collapse(I1.astype(np.uint8)))
        else:
            png(a, I0.astype(np.uint8), kind == "rgb")
            png(b, I1.astype(np.uint8), kind == "rgb")
        out = tmp_path / ("o_%s.flo" % kind)
        r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(a), str(b), str(out)] + args, capture_output=True, text=True)
        if ext == "png" and "libpng16" in r.stderr:
            require_or_skip(False, "libpng16 not present on this machine")
        assert r.returncode == 0, r.stderr
        flows[kind] = read_flo(out)
    assert np.array_equal(flows["gray"], flows["pgm"])
    assert np.array_equal(flows["rgb"], flows["pgm_collapsed"])


def test_brox_temporal_cli(synth, tmp_path):
    """brox_temporal nimages I1..In [...] dir verbose: dir/flowNN.flo byte-identical to the reference vectors"""
    c, g = CASES["broxt_seq4_64x48"], load("broxt_seq4_64x48")
    seq = synth.sequence(c["nx"], c["ny"], c["pair"])
    names = []
    for f in range(seq.shape[0]):
        names.append(str(tmp_path / ("f%d.pgm" % f)))
        write_pgm(names[-1], seq[f])
    p = c["params"]
    out = tmp_path / "flows"
    out.mkdir()
    r = subprocess.run([os.path.join(BIN, "brox_temporal"), str(seq.shape[0])] + names +
                       [str(p["alpha"]), str(p["gamma"]), str(p["nscales"]), str(p["nu"]), str(p["TOL"]), str(p["inner"]),
                        str(p["outer"]), str(out), "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    printed = [int(x.split("Iterations: ")[1]) for x in r.stdout.splitlines() if x.startswith("Iterations:")]
    assert printed == list(g["iters"]) and "Scale: 1" in r.stdout
    for f in range(seq.shape[0] - 1):
        want = np.stack([g["u"][f], g["v"][f]], axis=-1).astype(np.float32)
        assert np.array_equal(read_flo(out / ("flow%02d.flo" % f)), want)
    r = subprocess.run([os.path.join(BIN, "brox_temporal"), "2", names[0], names[1]], capture_output=True, text=True)
    assert r.returncode == 0 and "more than two frames" in r.stderr
    r = subprocess.run([os.path.join(BIN, "brox_temporal")], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage:" in r.stdout


def test_cli_option_table_warnings_and_substitutes(synth, tmp_path):
    """the table-driven option parser (cli/ofx_cli_common.h): tvl1flow warns -- only with verbose -- in the reference's
    order and formats (src/tvl1flow_main.cpp:102-167); horn_schunck_pyramidal turns zoom_factor >= 1 into 0.99, not the
    default (src/horn_schunck_pyramidal_main.cpp:109-112)"""
    nx, ny = 64, 48
    I0, I1 = synth.pair("P0", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    a, b = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), a, b, str(tmp_path / "o.flo"), "-3", "9", "-1", "0", "-5", "1.5", "0",
                        "-1", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    warn = [ln for ln in r.stderr.splitlines() if ln.startswith("warning:")]
    assert warn == ["warning: nproc changed to 0", "warning: tau changed to 0.25", "warning: lambda changed to 0.15",
                    "warning: theta changed to 0.3", "warning: nscales changed to 100", "warning: zfactor changed to 0.5",
                    "warning: nwarps changed to 5", "warning: epsilon changed to 0.010000"]
    quiet = subprocess.run([os.path.join(BIN, "tvl1flow"), a, b, str(tmp_path / "q.flo"), "-3", "9"], capture_output=True, text=True)
    assert quiet.returncode == 0 and "warning" not in quiet.stderr
    r = subprocess.run([os.path.join(BIN, "horn_schunck_pyramidal"), a, b, str(tmp_path / "h.flo"), "0", "-2", "2", "1.5", "0",
                        "-1", "20", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "alpha=7 nscales=2 zfactor=0.99 warps=10 epsilon=0.0001" in r.stderr


def test_tvl1flow_cli_many_scales(orc, synth, tmp_path):
    """zfactor = 0.9: the reference's rule gives 1 + log(hypot(320, 240) / 16) / log(1 / 0.9) = 31.6 -> 31 pyramid levels here
    (47 at 1080p, more than the 32 levels ofx_stats itemises): no fixed cap on the levels, .flo byte-identical"""
    nx, ny = 320, 240
    I0, I1 = synth.pair("P1", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    out = tmp_path / "o.flo"
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(out),
                        "0", "0.25", "0.15", "0.3", "100", "0.9", "2", "0.01", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "nscales=31 " in r.stderr
    uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=31, zfactor=0.9, warps=2)
    assert np.array_equal(read_flo(out), np.stack([uo, vo], axis=-1).astype(np.float32))
    # 34 levels (> OFX_MAX_SCALES = 32): zfactor = 0.91 -> 1 + log(400 / 16) / log(1 / 0.91) = 35.1 -> 35
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(out),
                        "0", "0.25", "0.15", "0.3", "100", "0.91", "1", "0.01", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "nscales=35 " in r.stderr
    uo, vo, _, _ = orc.tvl1_multiscale(I0, I1, nscales=35, zfactor=0.91, warps=1)
    assert np.array_equal(read_flo(out), np.stack([uo, vo], axis=-1).astype(np.float32))


def test_tvl1occflow_cli_is_drop_in(orc, synth, tmp_path):
    """three frames in, .flo and a 0 / 255 occlusion map out: the flow payload equals the oracle's cast to float, the map its
    thresholded chi; defaults, the auto-nscales rule (16-pixel floor on min(nx, ny)) and the unconditional warnings of
    src/tvl1occflow_main.cpp"""
    nx, ny = 96, 80
    seq = synth.sequence(nx, ny, 3, 1)
    names = []
    for k in range(3):
        write_pgm(tmp_path / ("f%d.pgm" % k), seq[k])
        names.append(str(tmp_path / ("f%d.pgm" % k)))
    frames = [s.astype(np.uint8).astype(np.float64) for s in seq]            # what write_pgm stores
    exe = os.path.join(BIN, "tvl1occflow")
    out, occ = tmp_path / "o.flo", tmp_path / "occ.pgm"
    #            I_1 I0 I1 I0_Smoothed out outOcc nproc lambda alpha beta theta nscales zfactor nwarps epsilon verbose
    r = subprocess.run([exe] + names + [names[1], str(out), str(occ), "1", "0.15", "0.01", "0.15", "0.3", "100", "0.5", "2",
                                        "0.01", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # N = floor(log(80 / 16) / log(2)) + 1 = 3 (:209-213)
    assert " nscales=3 " in r.stderr and "Scale 2: 24x20" in r.stderr
    uo, vo, co, it = orc.tvl1occ_multiscale(frames[0], frames[1], frames[2], nscales=3, warps=2)
    assert np.array_equal(read_flo(out), np.stack([uo, vo], axis=-1).astype(np.float32))
    raw = open(occ, "rb").read()
    head = b"P2\n96 80\n255\n"                                  # 7680 pixels <= 10000: ASCII (iio.cpp:3842-3849)
    assert raw.startswith(head)
    got = np.array(raw[len(head):].split(), dtype=np.int64).reshape(ny, nx)
    assert np.array_equal(got, (co * 255).astype(np.int64)) and 0 < got.mean() < 255
    printed = [int(x.split("Iterations: ")[1].split(",")[0]) for x in r.stderr.splitlines() if "Iterations:" in x]
    assert printed == list(it[::-1].ravel())
    # defaults (no I0_Smoothed -> I0; occlusions.png in the working directory); warnings do not wait for `verbose`, theta's does
    r2 = subprocess.run([exe] + names + [names[1], str(tmp_path / "o2.flo"), str(tmp_path / "occ2.pgm"), "-1", "0", "-2", "0",
                                         "0", "-4", "1.5", "0", "0"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    for word in ("nproc changed to 1", "lambda changed to 0.15", "alpha changed to 0.01", "beta changed to 0.15",
                 "nscales changed to 100", "zfactor changed to 0.5", "nwarps changed to 2", "epsilon changed to 0.010000"):
        assert "warning: " + word in r2.stderr, (word, r2.stderr)
    assert "theta changed" not in r2.stderr
    assert np.array_equal(read_flo(tmp_path / "o2.flo"), read_flo(out))
    assert open(tmp_path / "occ2.pgm", "rb").read() == raw
    r3 = subprocess.run([exe] + names, capture_output=True, text=True, cwd=tmp_path)
    assert r3.returncode == 0 or "libpng16" in r3.stderr, r3.stderr
    assert np.array_equal(read_flo(tmp_path / "flow.flo"), read_flo(out))
    if "libpng16" not in r3.stderr:
        assert open(tmp_path / "occlusions.png", "rb").read()[:8] == b"\x89PNG\r\n\x1a\n"
    # usage / size mismatch (the reference computes nothing and exits 0)
    assert subprocess.run([exe, names[0], names[1]], capture_output=True).returncode != 0
    write_pgm(tmp_path / "small.pgm", seq[0][:40, :40])
    r4 = subprocess.run([exe, names[0], names[1], str(tmp_path / "small.pgm"), names[1], str(tmp_path / "no.flo")], capture_output=True)
    assert r4.returncode == 0 and not (tmp_path / "no.flo").exists()


def test_front_end_stats_json(orc, synth, tmp_path):
    """OFX_STATS=path: the work record of the solve as JSON (per-scale sizes, iterations and error per warp -- what the
    reference prints as text when verbose, src/tvl1flow.cpp:184-188 -- plus iteration-kernel milliseconds); OFX_TOLERANCE=1
    switches the front-end to the f64 tolerance mode (AEPE bar, not byte identity)."""
    import json
    nx, ny = 160, 120
    I0, I1 = synth.pair("P1", nx, ny)
    write_pgm(tmp_path / "a.pgm", I0)
    write_pgm(tmp_path / "b.pgm", I1)
    env = dict(os.environ, OFX_STATS=str(tmp_path / "stats.json"))
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(tmp_path / "o.flo"),
                        "0", "0.25", "0.15", "0.3", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    d = json.load(open(tmp_path / "stats.json"))
    uo, vo, it, err = orc.tvl1_multiscale(I0, I1, nscales=4)
    assert d["program"] == "tvl1flow" and d["nscales"] == 4 and d["solves_per_scale"] == 5
    assert [s["iterations"] for s in d["scales"]] == [list(map(int, row)) for row in it]
    assert [(s["nx"], s["ny"]) for s in d["scales"]] == [(160, 120), (80, 60), (40, 30), (20, 15)]
    assert np.allclose([s["error"] for s in d["scales"]], err, rtol=1e-10, atol=0)
    assert d["work_pix_iters"] == sum(int(it[s].sum()) * d["scales"][s]["nx"] * d["scales"][s]["ny"] for s in range(4))
    assert all(s["iteration_kernel_ms"] > 0 for s in d["scales"]) and d["total_ms"] > 0
    want = np.stack([uo, vo], axis=-1).astype(np.float32)
    assert np.array_equal(read_flo(tmp_path / "o.flo"), want)
    env = dict(os.environ, OFX_TOLERANCE="1")
    r = subprocess.run([os.path.join(BIN, "tvl1flow"), str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(tmp_path / "t.flo"),
                        "0", "0.25", "0.15", "0.3", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = read_flo(tmp_path / "t.flo").astype(np.float64)
    assert float(np.mean(np.hypot(got[..., 0] - uo, got[..., 1] - vo))) < 1e-4
    assert not os.path.exists(tmp_path / "none.json")
