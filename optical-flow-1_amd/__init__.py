"""optical-flow-1_amd -- Python host-side mirror of the libofx.so C ABI (include/ofx.h).

The package name is not a Python identifier; load it with
``importlib.import_module("optical-flow-1_amd")`` (tests/conftest.py and bench.py do).

`Ofx` wraps one `ofx_ctx` (one GPU + one HIP stream).  Method names, argument order and meaning
follow the reference library functions they replace (src/tvl1flow.h, operators.h,
bicubic_interpolation.h, zoom.h, utils.h); arrays are row-major float64 numpy arrays of shape
(ny, nx).  There is no CPU fallback: if libofx.so is missing or no gfx950 device is usable, creating
an `Ofx` raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFX_LIB_PATH") or os.path.join(HERE, "libofx.so")   # override: A/B builds only

F64, F32 = 0, 1
MAX_SCALES, MAX_SOLVES = 32, 64

_STATUS = {1: "invalid argument", 2: "GaussianSmooth: sigma too large", 3: "out of memory",
           4: "HIP runtime error", 5: "no usable gfx950 device"}


class OfxError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("ofx status %d (%s)%s" % (status, _STATUS.get(status, "?"), ": " + detail if detail else ""))


class Stats(C.Structure):
    _fields_ = [("nscales", C.c_int), ("nsolves", C.c_int),
                ("nx", C.c_int * MAX_SCALES), ("ny", C.c_int * MAX_SCALES),
                ("iters", (C.c_int * MAX_SOLVES) * MAX_SCALES),
                ("error", (C.c_double * MAX_SOLVES) * MAX_SCALES),
                ("iter_ms", C.c_double * MAX_SCALES),
                ("iter_launches", C.c_longlong * MAX_SCALES),
                ("work_pix_iters", C.c_double), ("total_ms", C.c_double),
                ("odd_stops", C.c_int), ("odd_stops_stored", C.c_int), ("fused", C.c_int * MAX_SCALES)]

    def iterations(self):
        return np.array([[self.iters[s][w] for w in range(min(self.nsolves, MAX_SOLVES))]
                         for s in range(self.nscales)])

    def errors(self):
        return np.array([[self.error[s][w] for w in range(min(self.nsolves, MAX_SOLVES))]
                         for s in range(self.nscales)])


def build(verbose=False):
    """Compile libofx.so for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout, out.stderr)
    if out.returncode:
        raise RuntimeError("libofx.so build failed")


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_vp = C.c_void_p
_i, _d = C.c_int, C.c_double


def lib():
    """The loaded libofx.so with argtypes set (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    sig = {
        "ofx_device_count": (_i, []),
        "ofx_ctx_create": (_i, [C.POINTER(_vp), _i, _i]),
        "ofx_ctx_destroy": (None, [_vp]),
        "ofx_strerror": (C.c_char_p, [_i]),
        "ofx_last_error": (C.c_char_p, [_vp]),
        "ofx_ctx_stream": (_vp, [_vp]),
        "ofx_ctx_precision": (_i, [_vp]),
        "ofx_ctx_synchronize": (_i, [_vp]),
        "ofx_set_option": (_i, [_vp, C.c_char_p, _d]),
        "ofx_get_stats": (_i, [_vp, C.POINTER(Stats)]),
        "ofx_divergence": (_i, [_vp, _dp, _dp, _dp, _i, _i]),
        "ofx_forward_gradient": (_i, [_vp, _dp, _dp, _dp, _i, _i]),
        "ofx_centered_gradient": (_i, [_vp, _dp, _dp, _dp, _i, _i]),
        "ofx_dxx": (_i, [_vp, _dp, _dp, _i, _i]),
        "ofx_dyy": (_i, [_vp, _dp, _dp, _i, _i]),
        "ofx_dxy": (_i, [_vp, _dp, _dp, _i, _i]),
        "ofx_gaussian": (_i, [_vp, _dp, _i, _i, _d]),
        "ofx_bicubic_at": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i, _i]),
        "ofx_hypot": (_i, [_vp, _dp, _dp, _dp, _i]),
        "ofx_robust_expo": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _d, _d, _d, _i, _d, _d, _i, _i, _i]),
        "ofx_bicubic_warp": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i]),
        "ofx_zoom_size": (None, [_i, _i, C.POINTER(_i), C.POINTER(_i), _d]),
        "ofx_zoom_out": (_i, [_vp, _dp, _dp, _i, _i, _d]),
        "ofx_zoom_in": (_i, [_vp, _dp, _dp, _i, _i, _i, _i]),
        "ofx_image_normalization_2": (_i, [_vp, _dp, _dp, _dp, _dp, _i]),
        "ofx_image_normalization_1": (_i, [_vp, _dp, _dp, _i]),
        "ofx_getminmax": (_i, [_vp, _dp, _i, C.POINTER(_d), C.POINTER(_d)]),
        "ofx_centered_gradient3": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i]),
        "ofx_bicubic_at_color": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _i, _i]),
        "ofx_zoom_out_color": (_i, [_vp, _dp, _dp, _i, _i, _i, _d]),
        "ofx_tvl1_single_scale": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _d, _i, _d, _i]),
        "ofx_tvl1_multiscale": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _d, _i, _d, _i, _d, _i]),
        "ofx_tvl1_multiscale_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _d, _d, _d, _i, _d, _i, _d, _i]),
        "ofx_tvl1_group_dev": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _d, _d, _d, _i, _d, _i, _d,
                                    C.POINTER(Stats)]),
        "ofx_tvl1_batch_dev": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _d, _d, _d,
                                    _i, _d, _i, _d, C.POINTER(_d)]),
        "ofx_tvl1_batch_group_size": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _d]),
        "ofx_tvl1_iterations": (_i, [_vp] + [_dp] * 9 + [_i, _i, _d, _d, _d, _i, C.POINTER(_d)]),
        "ofx_hs_single_scale": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _d, _i, _d, _i, _i]),
        "ofx_hs_pyramidal": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _d, _i, _d, _i, _d, _i, _i]),
        "ofx_brox_spatial": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _i, _d, _d, _i, _i, _i]),
        "ofx_hs_group_dev": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _d, _i, _d, _i, _d, _i,
                                  C.POINTER(Stats)]),
        "ofx_hs_batch_dev": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _d, _i, _d,
                                  _i, _d, _i, C.POINTER(_d)]),
        "ofx_brox_group_dev": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _d, _d, _i, _d, _d, _i,
                                    _i, C.POINTER(Stats)]),
        "ofx_brox_batch_dev": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _d, _d, _i,
                                    _d, _d, _i, _i, C.POINTER(_d)]),
        "ofx_bicubic_warp_color": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i, _i]),
        "ofx_image_normalization_2_color": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i]),
        "ofx_image_normalization_3": (_i, [_vp, _dp, _dp, _dp, _i]),
        "ofx_image_normalization_4": (_i, [_vp] + [_dp] * 8 + [_i]),
        "ofx_me_median_filtering": (_i, [_vp, _dp, _i, _i, _i]),
        "ofx_solver_wrt_v": (_i, [_vp] + [_dp] * 17 + [_d, _d, _d, _i, _i]),
        "ofx_scalar_rof_box_cell_centered": (_i, [_vp, _dp, _dp, _dp, _dp, _dp, _d, _d, _i, _i, _i]),
        "ofx_solver_wrt_u": (_i, [_vp] + [_dp] * 6 + [_d, _d, _i, _i] + [_dp] * 4 + [_i]),
        "ofx_solver_wrt_chi": (_i, [_vp] + [_dp] * 14 + [_d] * 6 + [_i, _i, _dp, _dp, _i]),
        "ofx_tvl1occ_multiscale": (_i, [_vp] + [_dp] * 7 + [_i, _i, _d, _d, _d, _d, _i, _d, _i, _d, _i]),
        "ofx_tvl1occ_batch": (_i, [_vp, _i, _i] + [_vp] * 7 + [_i, _i, _d, _d, _d, _d, _i, _d, _i, _d]),
        "ofx_hs_classic": (_i, [_vp, _dp, _dp, _dp, _dp, _i, _i, _i, _d]),
        "ofx_brox_temporal": (_i, [_vp, _dp, _dp, _dp, _i, _i, _i, _d, _d, _i, _d, _d, _i, _i, _i]),
    }
    L.ofx_missing = []
    for name, (res, args) in sig.items():
        if not hasattr(L, name):          # stale / incomplete build: tests/test_abi.py fails on this list
            L.ofx_missing.append(name)
            continue
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def zoom_size(nx, ny, factor):
    a, b = _i(), _i()
    lib().ofx_zoom_size(nx, ny, C.byref(a), C.byref(b), factor)
    return a.value, b.value


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def tvl1_batch_dev(ctxs, dI0, dI1, d_flo, nx, ny, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5,
                   epsilon=0.01):
    """ofx_tvl1_batch_dev: lists of device pointers (ints), one entry per pair; the pairs are cut into lockstep
    groups of tvl1_batch_group_size() pairs, group q runs on ctxs[q % len(ctxs)] with len(ctxs) groups in flight.
    Returns the per-pair work (pixel-iterations)."""
    n = len(dI0)
    arr = lambda xs: (_vp * len(xs))(*xs)
    work = (_d * max(n, 1))()
    s = lib().ofx_tvl1_batch_dev(arr([c.h.value for c in ctxs]), len(ctxs), arr(dI0), arr(dI1), arr(d_flo), n, nx, ny, tau,
                                 lam, theta, nscales, zfactor, warps, epsilon, work)
    if s:
        raise OfxError(s, "; ".join((c.L.ofx_last_error(c.h) or b"").decode() for c in ctxs))
    return [work[i] for i in range(n)]


def _batch_call(fn, ctxs, dA, dB, d_flo, *args):
    n = len(dA)
    arr = lambda xs: (_vp * max(len(xs), 1))(*xs)
    work = (_d * max(n, 1))()
    s = fn(arr([c.h.value for c in ctxs]), len(ctxs), arr(dA), arr(dB), arr(d_flo), n, *args, work)
    if s:
        raise OfxError(s, "; ".join((c.L.ofx_last_error(c.h) or b"").decode() for c in ctxs))
    return [work[i] for i in range(n)]


def hs_batch_dev(ctxs, dI1, dI2, d_flo, nx, ny, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150):
    """ofx_hs_batch_dev: lists of device pointers (ints), one entry per pair; lockstep groups on ctxs[q % len(ctxs)].
    Returns the per-pair work (pixel-sweeps)."""
    return _batch_call(lib().ofx_hs_batch_dev, ctxs, dI1, dI2, d_flo, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter)


def brox_batch_dev(ctxs, dI1, dI2, d_flo, nx, ny, alpha=50.0, gamma=10.0, nscales=10, nu=0.5, TOL=1e-4, inner=1, outer=15):
    """ofx_brox_batch_dev (see hs_batch_dev)"""
    return _batch_call(lib().ofx_brox_batch_dev, ctxs, dI1, dI2, d_flo, nx, ny, alpha, gamma, nscales, nu, TOL, inner, outer)


def tvl1occ_batch(ctxs, triples, lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=3, zfactor=0.5, warps=2, epsilon=0.01, out=None):
    """ofx_tvl1occ_batch: triples = list of (I_1, I0, I1) or (I_1, I0, I1, filtI0) host images; lockstep groups of up to 16
    consecutive triples, group q on ctxs[q % len(ctxs)] (one host thread per context inside the library).  Returns a list of
    (u1, u2, chi); out = such a list from an earlier call: its planes are reused (fresh planes are first touched -- page
    faults -- while the library writes them, which a benchmark loop would otherwise time)."""
    n = len(triples)
    ny, nx = triples[0][1].shape
    ins = [[_f64(t[k] if k < len(t) else t[1]) for t in triples] for k in range(4)]
    outs = [[o[k] for o in out] for k in range(3)] if out is not None else [[np.empty((ny, nx)) for _ in range(n)] for _ in range(3)]
    ptr = lambda arrs: (_vp * n)(*[a.ctypes.data for a in arrs])
    s = lib().ofx_tvl1occ_batch((_vp * len(ctxs))(*[c.h.value for c in ctxs]), len(ctxs), n, *[ptr(a) for a in ins],
                                *[ptr(a) for a in outs], nx, ny, lam, alpha, beta, theta, nscales, zfactor, warps, epsilon)
    if s:
        raise OfxError(s, "; ".join((c.L.ofx_last_error(c.h) or b"").decode() for c in ctxs))
    return list(zip(*outs))


def tvl1_batch_group_size(ctxs, n_pairs, nx, ny, nscales=5, zfactor=0.5):
    """ofx_tvl1_batch_group_size: pairs per lockstep group tvl1_batch_dev uses for a batch of n_pairs."""
    g = lib().ofx_tvl1_batch_group_size((_vp * len(ctxs))(*[c.h.value for c in ctxs]), len(ctxs), n_pairs, nx, ny, nscales,
                                        zfactor)
    if g < 1:
        raise OfxError(g, "; ".join((c.L.ofx_last_error(c.h) or b"").decode() for c in ctxs))
    return g


class Ofx:
    """One libofx context.  precision: F64 (strict) or F32 (fast, float storage)."""

    def __init__(self, device=0, precision=F64):
        self.L = lib()
        h = _vp()
        s = self.L.ofx_ctx_create(C.byref(h), device, precision)
        if s:
            raise OfxError(s, "ofx_ctx_create(device=%d)" % device)
        self.h = h
        self.precision = precision

    def close(self):
        if getattr(self, "h", None):
            self.L.ofx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, s):
        if s:
            raise OfxError(s, (self.L.ofx_last_error(self.h) or b"").decode())

    def set_option(self, name, value):
        self._ck(self.L.ofx_set_option(self.h, name.encode(), float(value)))

    def stats(self):
        st = Stats()
        self._ck(self.L.ofx_get_stats(self.h, C.byref(st)))
        return st

    def stream(self):
        return self.L.ofx_ctx_stream(self.h)

    def synchronize(self):
        self._ck(self.L.ofx_ctx_synchronize(self.h))

    # ---- operators ---------------------------------------------------------------------------
    def divergence(self, v1, v2):
        ny, nx = v1.shape
        out = np.empty((ny, nx))
        self._ck(self.L.ofx_divergence(self.h, _f64(v1), _f64(v2), out, nx, ny))
        return out

    def forward_gradient(self, f):
        ny, nx = f.shape
        fx, fy = np.empty((ny, nx)), np.empty((ny, nx))
        self._ck(self.L.ofx_forward_gradient(self.h, _f64(f), fx, fy, nx, ny))
        return fx, fy

    def centered_gradient(self, f):
        ny, nx = f.shape
        fx, fy = np.empty((ny, nx)), np.empty((ny, nx))
        self._ck(self.L.ofx_centered_gradient(self.h, _f64(f), fx, fy, nx, ny))
        return fx, fy

    def _second(self, fn, f):
        ny, nx = f.shape
        out = np.empty((ny, nx))
        self._ck(fn(self.h, _f64(f), out, nx, ny))
        return out

    def dxx(self, f): return self._second(self.L.ofx_dxx, f)
    def dyy(self, f): return self._second(self.L.ofx_dyy, f)
    def dxy(self, f): return self._second(self.L.ofx_dxy, f)

    def gaussian(self, I, sigma):
        ny, nx = I.shape
        out = _f64(I).copy()
        self._ck(self.L.ofx_gaussian(self.h, out, nx, ny, sigma))
        return out

    def bicubic_at(self, I, uu, vv, border_out=False):
        ny, nx = I.shape
        uu, vv = _f64(np.atleast_1d(uu)), _f64(np.atleast_1d(vv))
        out = np.empty(uu.shape)
        self._ck(self.L.ofx_bicubic_at(self.h, _f64(I), uu, vv, out, uu.size, nx, ny, int(border_out)))
        return out

    def robust_expo(self, I1, I2, method=1, alpha=50.0, gamma=10.0, lam=1.0, nscales=5, nu=0.5, TOL=1e-4, inner=1, outer=15,
                    verbose=0, nz=1):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        self._ck(self.L.ofx_robust_expo(self.h, _f64(I1), _f64(I2), u, v, nx, ny, nz, method, alpha, gamma, lam, nscales, nu, TOL,
                                        inner, outer, verbose))
        return u, v

    def hypot(self, x, y):
        x, y = _f64(np.atleast_1d(x)).ravel(), _f64(np.atleast_1d(y)).ravel()
        out = np.empty(x.shape)
        self._ck(self.L.ofx_hypot(self.h, x, y, out, x.size))
        return out

    def bicubic_warp(self, I, u, v, border_out=False):
        ny, nx = I.shape
        out = np.empty((ny, nx))
        self._ck(self.L.ofx_bicubic_warp(self.h, _f64(I), _f64(u), _f64(v), out, nx, ny, int(border_out)))
        return out

    def zoom_out(self, I, factor):
        ny, nx = I.shape
        nxx, nyy = zoom_size(nx, ny, factor)
        out = np.empty((nyy, nxx))
        self._ck(self.L.ofx_zoom_out(self.h, _f64(I), out, nx, ny, factor))
        return out

    def zoom_in(self, I, nxx, nyy):
        ny, nx = I.shape
        out = np.empty((nyy, nxx))
        self._ck(self.L.ofx_zoom_in(self.h, _f64(I), out, nx, ny, nxx, nyy))
        return out

    def image_normalization_2(self, I1, I2):
        a, b = np.empty(I1.shape), np.empty(I2.shape)
        self._ck(self.L.ofx_image_normalization_2(self.h, _f64(I1), _f64(I2), a, b, I1.size))
        return a, b

    # ---- TV-L1 ---------------------------------------------------------------------------------
    def tvl1_single_scale(self, I0, I1, u1, u2, tau=0.25, lam=0.15, theta=0.3, warps=5, epsilon=0.01, verbose=0):
        ny, nx = I0.shape
        u1, u2 = _f64(u1).copy(), _f64(u2).copy()
        self._ck(self.L.ofx_tvl1_single_scale(self.h, _f64(I0), _f64(I1), u1, u2, nx, ny, tau, lam, theta, warps,
                                              epsilon, verbose))
        return u1, u2

    def tvl1_multiscale(self, I0, I1, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5, epsilon=0.01,
                        verbose=0, out=None):
        """out = (u1, u2): the caller's own (ny, nx) float64 planes to fill -- a loop over frames reuses them; fresh arrays
        cost their page faults inside the call (the download is their first touch)"""
        ny, nx = I0.shape
        u1, u2 = out if out is not None else (np.zeros((ny, nx)), np.zeros((ny, nx)))
        self._ck(self.L.ofx_tvl1_multiscale(self.h, _f64(I0), _f64(I1), u1, u2, nx, ny, tau, lam, theta, nscales,
                                            zfactor, warps, epsilon, verbose))
        return u1, u2

    def tvl1_multiscale_dev(self, dI0, dI1, d_flo, nx, ny, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5,
                            warps=5, epsilon=0.01, verbose=0):
        """dI0, dI1, d_flo: device pointers (ints).  Enqueues on the context's stream."""
        self._ck(self.L.ofx_tvl1_multiscale_dev(self.h, dI0, dI1, d_flo, nx, ny, tau, lam, theta, nscales, zfactor,
                                                warps, epsilon, verbose))

    def tvl1_group_dev(self, dI0, dI1, d_flo, nx, ny, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5,
                       epsilon=0.01):
        """ofx_tvl1_group_dev: lists of device pointers (ints), all pairs solved in lockstep on this context.
        Returns the per-pair Stats records."""
        n = len(dI0)
        arr = lambda xs: (_vp * len(xs))(*xs)
        st = (Stats * max(n, 1))()
        self._ck(self.L.ofx_tvl1_group_dev(self.h, n, arr(dI0), arr(dI1), arr(d_flo), nx, ny, tau, lam, theta, nscales,
                                           zfactor, warps, epsilon, st))
        return [st[i] for i in range(n)]

    def _group_call(self, fn, dA, dB, d_flo, *args):
        n = len(dA)
        arr = lambda xs: (_vp * max(len(xs), 1))(*xs)
        st = (Stats * max(n, 1))()
        self._ck(fn(self.h, n, arr(dA), arr(dB), arr(d_flo), *args, st))
        return [st[i] for i in range(n)]

    def hs_group_dev(self, dI1, dI2, d_flo, nx, ny, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150):
        """ofx_hs_group_dev: all pairs solved in lockstep on this context; returns the per-pair Stats records"""
        return self._group_call(self.L.ofx_hs_group_dev, dI1, dI2, d_flo, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter)

    def brox_group_dev(self, dI1, dI2, d_flo, nx, ny, alpha=50.0, gamma=10.0, nscales=10, nu=0.5, TOL=1e-4, inner=1, outer=15):
        """ofx_brox_group_dev (see hs_group_dev)"""
        return self._group_call(self.L.ofx_brox_group_dev, dI1, dI2, d_flo, nx, ny, alpha, gamma, nscales, nu, TOL, inner, outer)

    def tvl1_iterations(self, u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, tau, lam, theta, n_iter):
        """In place on the six state arrays (float64, C-contiguous); returns the last error."""
        ny, nx = u1.shape
        err = _d()
        self._ck(self.L.ofx_tvl1_iterations(self.h, u1, u2, p11, p12, p21, p22, _f64(I1wx), _f64(I1wy), _f64(rho_c),
                                            nx, ny, tau, lam, theta, n_iter, C.byref(err)))
        return err.value

    # ---- Horn-Schunck / Brox ---------------------------------------------------------------------
    def hs_single_scale(self, I1, I2, u, v, alpha=7.0, warps=10, TOL=1e-4, maxiter=150, verbose=0):
        ny, nx = I1.shape
        u, v = _f64(u).copy(), _f64(v).copy()
        self._ck(self.L.ofx_hs_single_scale(self.h, _f64(I1), _f64(I2), u, v, nx, ny, alpha, warps, TOL, maxiter, verbose))
        return u, v

    def hs_pyramidal(self, I1, I2, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150, verbose=0, out=None):
        """out = (u, v) planes to reuse (the library overwrites them; fresh planes are page-faulted while it does)"""
        ny, nx = I1.shape
        u, v = out if out is not None else (np.zeros((ny, nx)), np.zeros((ny, nx)))
        self._ck(self.L.ofx_hs_pyramidal(self.h, _f64(I1), _f64(I2), u, v, nx, ny, alpha, nscales, zfactor, warps, TOL,
                                         maxiter, verbose))
        return u, v

    def brox_spatial(self, I1, I2, alpha=50.0, gamma=10.0, nscales=10, nu=0.5, TOL=1e-4, inner=1, outer=15, verbose=0, out=None):
        ny, nx = I1.shape
        u, v = out if out is not None else (np.zeros((ny, nx)), np.zeros((ny, nx)))
        self._ck(self.L.ofx_brox_spatial(self.h, _f64(I1), _f64(I2), u, v, nx, ny, alpha, gamma, nscales, nu, TOL,
                                         inner, outer, verbose))
        return u, v

    def brox_temporal(self, I, alpha=18.0, gamma=7.0, nscales=10, nu=0.75, TOL=1e-4, inner=1, outer=15, verbose=0):
        """I: (frames, ny, nx) -> u, v of shape (frames - 1, ny, nx)"""
        frames, ny, nx = I.shape
        u, v = np.zeros((max(frames - 1, 0), ny, nx)), np.zeros((max(frames - 1, 0), ny, nx))
        self._ck(self.L.ofx_brox_temporal(self.h, _f64(I), u, v, nx, ny, frames, alpha, gamma, nscales, nu, TOL, inner,
                                          outer, verbose))
        return u, v

    # ---- colour / sequence variants of the operator surface --------------------------------------------------
    def centered_gradient3(self, f):
        nz, ny, nx = f.shape
        dx, dy, dz = np.empty((nz, ny, nx)), np.empty((nz, ny, nx)), np.empty((nz, ny, nx))
        self._ck(self.L.ofx_centered_gradient3(self.h, _f64(f), dx, dy, dz, nx, ny, nz))
        return dx, dy, dz

    def bicubic_at_color(self, I, uu, vv, k, border_out=False):
        """I: (ny, nx, nz) interleaved channels; uu, vv: arrays of sample coordinates"""
        ny, nx, nz = I.shape
        uu, vv = _f64(np.atleast_1d(uu)), _f64(np.atleast_1d(vv))
        out = np.empty(uu.shape)
        self._ck(self.L.ofx_bicubic_at_color(self.h, _f64(I), uu, vv, out, uu.size, nx, ny, nz, k, int(border_out)))
        return out

    def zoom_out_color(self, I, factor):
        ny, nx, nz = I.shape
        nxx, nyy = zoom_size(nx, ny, factor)
        out = np.empty((nyy, nxx, nz))
        self._ck(self.L.ofx_zoom_out_color(self.h, _f64(I), out, nx, ny, nz, factor))
        return out

    def image_normalization_1(self, I):
        out = np.empty(I.shape)
        self._ck(self.L.ofx_image_normalization_1(self.h, _f64(I), out, I.size))
        return out

    def getminmax(self, x):
        a, b = _d(), _d()
        self._ck(self.L.ofx_getminmax(self.h, _f64(x), x.size, C.byref(a), C.byref(b)))
        return a.value, b.value

    def hs_classic(self, a, b, niter, alpha):
        h, w = a.shape
        u, v = np.zeros((h, w)), np.zeros((h, w))
        self._ck(self.L.ofx_hs_classic(self.h, _f64(a), _f64(b), u, v, w, h, niter, alpha))
        return u, v

    # ---- SURVEY 8(f)4 colour operators / 8(f)1 building blocks of TV-L1 with occlusions ------------------------------
    def bicubic_warp_color(self, I, u, v, border_out=False):
        """I: (ny, nx, nz) interleaved channels; u, v: (ny, nx)"""
        ny, nx, nz = I.shape
        out = np.empty((ny, nx, nz))
        self._ck(self.L.ofx_bicubic_warp_color(self.h, _f64(I), _f64(u), _f64(v), out, nx, ny, nz, int(border_out)))
        return out

    def image_normalization_2_color(self, I1, I2, size=None):
        nz = I1.shape[-1]
        a, b = _f64(I1).copy(), _f64(I2).copy()
        self._ck(self.L.ofx_image_normalization_2_color(self.h, _f64(I1), _f64(I2), a, b, I1.size if size is None else size, nz))
        return a, b

    def image_normalization_3(self, I0, I1, I2):
        a, b, c = _f64(I0).copy(), _f64(I1).copy(), _f64(I2).copy()
        self._ck(self.L.ofx_image_normalization_3(self.h, a, b, c, a.size))
        return a, b, c

    def image_normalization_4(self, I_1, I0, I1, F):
        outs = [np.empty(I0.shape) for _ in range(4)]
        self._ck(self.L.ofx_image_normalization_4(self.h, _f64(I_1), _f64(I0), _f64(I1), _f64(F), *outs, I0.size))
        return outs

    def median_filtering(self, I, wsize=3):
        ny, nx = I.shape
        out = _f64(I).copy()
        self._ck(self.L.ofx_me_median_filtering(self.h, out, nx, ny, wsize))
        return out

    def occ_solver_v(self, u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, grad1, grad3, alpha, theta, lam):
        """Solver_wrt_v -> (v1, v2, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2)"""
        ny, nx = u1.shape
        v1, v2, f1, f2, b1, b2 = outs = [np.empty((ny, nx)) for _ in range(6)]
        self._ck(self.L.ofx_solver_wrt_v(self.h, _f64(u1), _f64(u2), v1, v2, _f64(chi), _f64(I1wx), _f64(I1wy), _f64(I_1wx),
                                         _f64(I_1wy), _f64(rho1_c), _f64(rho3_c), f1, f2, b1, b2, _f64(grad1), _f64(grad3),
                                         alpha, theta, lam, nx, ny))
        return outs

    def occ_solver_chi(self, u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vf1, Vf2, Vb1, Vb2, g, lam, theta, alpha,
                       beta, tau_chi, tau_eta, eta1=None, eta2=None, n_iter=100):
        """Solver_wrt_chi with the dual variable as explicit state -> (chi, eta1, eta2); eta defaults to zero"""
        ny, nx = u1.shape
        chi = _f64(chi).copy()
        eta1 = np.zeros((ny, nx)) if eta1 is None else _f64(eta1).copy()
        eta2 = np.zeros((ny, nx)) if eta2 is None else _f64(eta2).copy()
        self._ck(self.L.ofx_solver_wrt_chi(self.h, _f64(u1), _f64(u2), chi, _f64(I1wx), _f64(I1wy), _f64(I_1wx), _f64(I_1wy),
                                           _f64(rho1_c), _f64(rho3_c), _f64(Vf1), _f64(Vf2), _f64(Vb1), _f64(Vb2), _f64(g), lam,
                                           theta, alpha, beta, tau_chi, tau_eta, nx, ny, eta1, eta2, n_iter))
        return chi, eta1, eta2

    def rof_box(self, u, f, P1, P2, g, lam, omega, n_iter):
        """Scalar_ROF_BoxCellCentered -> (u, P1, P2)"""
        ny, nx = u.shape
        u, P1, P2 = _f64(u).copy(), _f64(P1).copy(), _f64(P2).copy()
        self._ck(self.L.ofx_scalar_rof_box_cell_centered(self.h, u, _f64(f), P1, P2, _f64(g), lam, omega, nx, ny, n_iter))
        return u, P1, P2

    def tvl1occ_multiscale(self, I_1, I0, I1, filtI0=None, lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=3, zfactor=0.5,
                           warps=2, epsilon=0.01, verbose=0, out=None):
        """Dual_TVL1_optic_flow_multiscale with occlusions (src/tvl1occflow.h) -> (u1, u2, chi); stats() has the outer iterations.
        out = (u1, u2, chi): result planes to reuse (freshly allocated planes cost ~3 ms of page faults per 16 MB when the
        library writes them -- a measuring artefact of a benchmark loop, not of the solve)"""
        ny, nx = I0.shape
        filtI0 = I0 if filtI0 is None else filtI0
        u1, u2, chi = out if out is not None else (np.empty((ny, nx)), np.empty((ny, nx)), np.empty((ny, nx)))
        self._ck(self.L.ofx_tvl1occ_multiscale(self.h, _f64(I_1), _f64(I0), _f64(I1), _f64(filtI0), u1, u2, chi, nx, ny, lam, alpha,
                                               beta, theta, nscales, zfactor, warps, epsilon, verbose))
        return u1, u2, chi

    def occ_solver_u(self, v1, v2, chi, g, theta, beta, p=None, n_iter=10):
        """Solver_wrt_u with the four dual planes as explicit state -> (u1, u2, [p11, p12, p21, p22]); p defaults to zero"""
        ny, nx = v1.shape
        u1, u2 = np.empty((ny, nx)), np.empty((ny, nx))
        p = [np.zeros((ny, nx)) for _ in range(4)] if p is None else [_f64(a).copy() for a in p]
        self._ck(self.L.ofx_solver_wrt_u(self.h, u1, u2, _f64(v1), _f64(v2), _f64(chi), _f64(g), theta, beta, nx, ny, *p, n_iter))
        return u1, u2, p
