"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for the two CPU checkers:

* ``Oracle()``  -> oracle/liboracle.so, our own C restatement of the reference hot path
  (oracle/ofx_oracle.c, double storage + arithmetic, OpenMP pragmas as in the reference).
* ``Ref()``     -> oracle/_ref/libofref.so, the reference's own sources compiled by
  oracle/Makefile where they lie under /root/reference/src (binary only; git-ignored).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (optical-flow-1_amd, libofx.so) never does, and has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libofref.so")
REF32_SO = os.path.join(HERE, "_ref", "libofref32.so")      # the reference's float build (ofpix_t = float)

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = C.POINTER(C.c_int)


def build(verbose=False):
    """Compile liboracle.so (and _ref/libofref.so when /root/reference is present)."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout, out.stderr)
    if out.returncode:
        raise RuntimeError("oracle build failed")


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota
    (a GPU box hands one GPU's share of the host, not the whole machine) and OFX_CPU_THREADS."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    if os.environ.get("OFX_CPU_THREADS"):
        n = max(1, int(os.environ["OFX_CPU_THREADS"]))
    return n


def have_ref():
    return os.path.exists(REF_SO)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class _Lib:
    prefix = ""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle`)")
        self.lib = C.CDLL(path)

    def _fn(self, name, restype, *argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f

    # ---- operators ---------------------------------------------------------------------------
    def divergence(self, v1, v2):
        ny, nx = v1.shape
        out = np.empty((ny, nx))
        self._fn("divergence", None, _dp, _dp, _dp, C.c_int, C.c_int)(_f64(v1), _f64(v2), out, nx, ny)
        return out

    def forward_gradient(self, f):
        ny, nx = f.shape
        fx, fy = np.empty((ny, nx)), np.empty((ny, nx))
        self._fn("forward_gradient", None, _dp, _dp, _dp, C.c_int, C.c_int)(_f64(f), fx, fy, nx, ny)
        return fx, fy

    def centered_gradient(self, f):
        ny, nx = f.shape
        fx, fy = np.empty((ny, nx)), np.empty((ny, nx))
        self._fn("centered_gradient", None, _dp, _dp, _dp, C.c_int, C.c_int)(_f64(f), fx, fy, nx, ny)
        return fx, fy

    def _second(self, name, f):
        ny, nx = f.shape
        out = np.empty((ny, nx))
        self._fn(name, None, _dp, _dp, C.c_int, C.c_int)(_f64(f), out, nx, ny)
        return out

    def gaussian(self, I, sigma):
        ny, nx = I.shape
        out = _f64(I).copy()
        rc = self._fn("gaussian", C.c_int, _dp, C.c_int, C.c_int, C.c_double)(out, nx, ny, sigma)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return out

    def bicubic_at(self, I, uu, vv, border_out=False):
        ny, nx = I.shape
        return self._fn("bicubic_at", C.c_double, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            _f64(I), uu, vv, nx, ny, int(border_out))

    def bicubic_warp(self, I, u, v, border_out=False):
        ny, nx = I.shape
        out = np.empty((ny, nx))
        self._fn("bicubic_warp", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int)(
            _f64(I), _f64(u), _f64(v), out, nx, ny, int(border_out))
        return out

    def zoom_size(self, nx, ny, factor):
        a, b = C.c_int(), C.c_int()
        self._fn("zoom_size", None, C.c_int, C.c_int, _ip, _ip, C.c_double)(nx, ny, C.byref(a), C.byref(b), factor)
        return a.value, b.value

    def zoom_out(self, I, factor):
        ny, nx = I.shape
        nxx, nyy = self.zoom_size(nx, ny, factor)
        out = np.empty((nyy, nxx))
        rc = self._fn("zoom_out", C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double)(_f64(I), out, nx, ny, factor)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return out

    def zoom_in(self, I, nxx, nyy):
        ny, nx = I.shape
        out = np.empty((nyy, nxx))
        self._fn("zoom_in", None, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int)(_f64(I), out, nx, ny, nxx, nyy)
        return out

    def centered_gradient3(self, f):
        nz, ny, nx = f.shape
        dx, dy, dz = np.empty_like(f, dtype=np.float64), np.empty_like(f, dtype=np.float64), np.empty_like(f, dtype=np.float64)
        self._fn("centered_gradient3", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int)(_f64(f), dx, dy, dz, nx, ny, nz)
        return dx, dy, dz

    def bicubic_at_color(self, I, uu, vv, k, border_out=False):
        """I: (ny, nx, nz) interleaved channels"""
        ny, nx, nz = I.shape
        return self._fn("bicubic_at_color", C.c_double, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                        C.c_int)(_f64(I), uu, vv, nx, ny, nz, k, int(border_out))

    def getminmax(self, x):
        a, b = C.c_double(), C.c_double()
        self._fn("getminmax", None, C.POINTER(C.c_double), C.POINTER(C.c_double), _dp, C.c_int)(C.byref(a), C.byref(b), _f64(x), x.size)
        return a.value, b.value

    def hs_classic(self, a, b, niter, alpha):
        """src/horn_schunck_classic.cpp hs(): niter Jacobi iterations from a zero flow"""
        h, w = a.shape
        u, v = np.zeros((h, w)), np.zeros((h, w))
        self._fn("hs_classic", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double)(u, v, _f64(a), _f64(b), w, h, niter, alpha)
        return u, v

    def image_normalization_1(self, I):
        out = np.empty(I.shape)
        self._fn("image_normalization_1", None, _dp, _dp, C.c_int)(_f64(I), out, I.size)
        return out

    def image_normalization_2(self, I1, I2):
        a, b = np.empty(I1.shape), np.empty(I2.shape)
        self._fn("image_normalization_2", None, _dp, _dp, _dp, _dp, C.c_int)(_f64(I1), _f64(I2), a, b, I1.size)
        return a, b

    # ---- SURVEY 8(f)4 colour operators / 8(f)1 building blocks of TV-L1 with occlusions ----------------------------
    def bicubic_warp_color(self, I, u, v, border_out=False):
        """I: (ny, nx, nz) interleaved channels; u, v: (ny, nx)"""
        ny, nx, nz = I.shape
        out = np.empty((ny, nx, nz))
        self._fn("bicubic_warp_color", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int)(
            _f64(I), _f64(u), _f64(v), out, nx, ny, nz, int(border_out))
        return out

    def image_normalization_2_color(self, I1, I2, size=None):
        """I1, I2: (..., nz) interleaved; `size` = element count handed to the reference (default: all)"""
        nz = I1.shape[-1]
        a, b = _f64(I1).copy(), _f64(I2).copy()
        self._fn("image_normalization_2_color", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int)(
            _f64(I1), _f64(I2), a, b, I1.size if size is None else size, nz)
        return a, b

    def image_normalization_3(self, I0, I1, I2):
        a, b, c = _f64(I0).copy(), _f64(I1).copy(), _f64(I2).copy()
        self._fn("image_normalization_3", None, _dp, _dp, _dp, C.c_int)(a, b, c, a.size)
        return a, b, c

    def image_normalization_4(self, I_1, I0, I1, F):
        outs = [np.empty(I0.shape) for _ in range(4)]
        self._fn("image_normalization_4", None, *([_dp] * 8), C.c_int)(_f64(I_1), _f64(I0), _f64(I1), _f64(F), *outs, I0.size)
        return outs

    def median_filtering(self, I, wsize=3):
        ny, nx = I.shape
        out = _f64(I).copy()
        self._fn("median_filtering", None, _dp, C.c_int, C.c_int, C.c_int)(out, nx, ny, wsize)
        return out

    def rof_box(self, u, f, P1, P2, g, lam, omega, n_iter):
        """Scalar_ROF_BoxCellCentered -> (u, P1, P2)"""
        ny, nx = u.shape
        u, P1, P2 = _f64(u).copy(), _f64(P1).copy(), _f64(P2).copy()
        self._fn("rof_box", None, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            u, _f64(f), P1, P2, _f64(g), lam, omega, nx, ny, n_iter)
        return u, P1, P2

    def occ_solver_v(self, u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, grad1, grad3, alpha, theta, lam):
        """Solver_wrt_v -> (v1, v2, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2)"""
        ny, nx = u1.shape
        outs = [np.empty((ny, nx)) for _ in range(6)]
        v1, v2, f1, f2, b1, b2 = outs
        self._fn("occ_solver_v", None, *([_dp] * 17), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int)(
            _f64(u1), _f64(u2), v1, v2, _f64(chi), _f64(I1wx), _f64(I1wy), _f64(I_1wx), _f64(I_1wy), _f64(rho1_c),
            _f64(rho3_c), f1, f2, b1, b2, _f64(grad1), _f64(grad3), alpha, theta, lam, nx, ny)
        return outs


class Oracle(_Lib):
    """Our C restatement (oracle/ofx_oracle.c)."""
    prefix = "orc_"
    kind = "port"

    def tvl1occ_multiscale(self, I_1, I0, I1, filtI0=None, lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=3, zfactor=0.5,
                           warps=2, epsilon=0.01, verbose=0):
        """TV-L1 with occlusions (zero-filled-heap semantics) -> (u1, u2, chi, outer-iteration table [scale][warp])"""
        ny, nx = I0.shape
        filtI0 = I0 if filtI0 is None else filtI0
        u1, u2, chi = np.zeros((ny, nx)), np.zeros((ny, nx)), np.zeros((ny, nx))
        iters = (C.c_int * (warps * nscales))()
        rc = self._fn("tvl1occ_multiscale", C.c_int, *([_dp] * 7), C.c_int, C.c_int, *([C.c_double] * 4), C.c_int, C.c_double,
                      C.c_int, C.c_double, C.c_int, _ip)(_f64(I_1), _f64(I0), _f64(I1), _f64(filtI0), u1, u2, chi, nx, ny, lam,
                                                          alpha, beta, theta, nscales, zfactor, warps, epsilon, verbose, iters)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u1, u2, chi, np.array(list(iters)).reshape(nscales, warps)

    def occ_solver_u(self, v1, v2, chi, g, theta, beta, p=None, n_iter=10):
        """Solver_wrt_u with the four dual planes as explicit state -> (u1, u2, [p11, p12, p21, p22]); p defaults to zero"""
        ny, nx = v1.shape
        u1, u2 = np.empty((ny, nx)), np.empty((ny, nx))
        p = [np.zeros((ny, nx)) for _ in range(4)] if p is None else [_f64(a).copy() for a in p]
        self._fn("occ_solver_u", None, *([_dp] * 6), C.c_double, C.c_double, C.c_int, C.c_int, *([_dp] * 4), C.c_int)(
            u1, u2, _f64(v1), _f64(v2), _f64(chi), _f64(g), theta, beta, nx, ny, *p, n_iter)
        return u1, u2, p

    def occ_solver_chi(self, u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vf1, Vf2, Vb1, Vb2, g, lam, theta, alpha,
                       beta, tau_chi, tau_eta, eta1=None, eta2=None, n_iter=100):
        """Solver_wrt_chi with the dual variable as explicit state -> (chi, eta1, eta2); eta defaults to zero"""
        ny, nx = u1.shape
        chi = _f64(chi).copy()
        eta1 = np.zeros((ny, nx)) if eta1 is None else _f64(eta1).copy()
        eta2 = np.zeros((ny, nx)) if eta2 is None else _f64(eta2).copy()
        self._fn("occ_solver_chi", None, *([_dp] * 14), *([C.c_double] * 6), C.c_int, C.c_int, _dp, _dp, C.c_int)(
            _f64(u1), _f64(u2), chi, _f64(I1wx), _f64(I1wy), _f64(I_1wx), _f64(I_1wy), _f64(rho1_c), _f64(rho3_c), _f64(Vf1),
            _f64(Vf2), _f64(Vb1), _f64(Vb2), _f64(g), lam, theta, alpha, beta, tau_chi, tau_eta, nx, ny, eta1, eta2, n_iter)
        return chi, eta1, eta2

    def __init__(self):
        super().__init__(ORACLE_SO)

    def set_num_threads(self, n):
        self._fn("set_num_threads", None, C.c_int)(n)

    def max_threads(self):
        return self._fn("max_threads", C.c_int)()

    def dxx(self, f): return self._second("dxx", f)
    def dyy(self, f): return self._second("dyy", f)
    def dxy(self, f): return self._second("dxy", f)

    def set_sor_order(self, order):
        """0 = reference (lexicographic) sweeps, 1 = the HIP path's colour order (checker aid)."""
        self._fn("set_sor_order", None, C.c_int)(order)

    def set_sor_colour_levels(self, mask):
        """With order 1: bit s set = pyramid level s in colour order, clear = reference order (the HIP option of that name)."""
        self._fn("set_sor_colour_levels", None, C.c_uint)(mask & 0xFFFFFFFF)

    def set_sor_tile(self, w, h):
        """Brox order 3: tile size of the checkerboard sweeps (the HIP path: 64 rows x option sor_tile_w columns)."""
        self._fn("set_sor_tile", None, C.c_int, C.c_int)(w, h)

    def set_sor_wave_levels(self, n):
        """With order 1: Brox levels 0 .. n - 1 are swept as a checkerboard of tiles (order 3) -- the HIP option of that name."""
        self._fn("set_sor_wave_levels", None, C.c_int)(n)

    def set_sor_exact_tail(self, tail):
        """With order 1: the last `tail` solves of the finest level keep the reference's order (the HIP option of that name)."""
        self._fn("set_sor_exact_tail", None, C.c_int)(tail)

    def tvl1_single_scale(self, I0, I1, u1, u2, tau=0.25, lam=0.15, theta=0.3, warps=5, epsilon=0.01,
                          verbose=0):
        ny, nx = I0.shape
        u1, u2 = _f64(u1).copy(), _f64(u2).copy()
        iters = (C.c_int * warps)()
        errs = np.zeros(warps)
        self._fn("tvl1_single_scale", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                 C.c_double, C.c_int, C.c_double, C.c_int, _ip, _dp)(
            _f64(I0), _f64(I1), u1, u2, nx, ny, tau, lam, theta, warps, epsilon, verbose, iters, errs)
        return u1, u2, list(iters), errs

    def tvl1_multiscale(self, I0, I1, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5,
                        epsilon=0.01, verbose=0):
        ny, nx = I0.shape
        u1, u2 = np.zeros((ny, nx)), np.zeros((ny, nx))
        iters = (C.c_int * (warps * nscales))()
        errs = np.zeros(warps * nscales)
        rc = self._fn("tvl1_multiscale", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double,
                      C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, _ip, _dp)(
            _f64(I0), _f64(I1), u1, u2, nx, ny, tau, lam, theta, nscales, zfactor, warps, epsilon, verbose,
            iters, errs)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u1, u2, np.array(list(iters)).reshape(nscales, warps), errs.reshape(nscales, warps)

    def tvl1_iterations(self, u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, grad, tau, lam, theta, n_iter):
        """In place on the six state arrays; returns the last error."""
        ny, nx = u1.shape
        return self._fn("tvl1_iterations", C.c_double, *([_dp] * 10), C.c_int, C.c_int, C.c_double, C.c_double,
                        C.c_double, C.c_int)(u1, u2, p11, p12, p21, p22, _f64(I1wx), _f64(I1wy), _f64(rho_c),
                                             _f64(grad), nx, ny, tau, lam, theta, n_iter)

    def hs_single_scale(self, I1, I2, u, v, alpha=7.0, warps=10, TOL=1e-4, maxiter=150, verbose=0):
        ny, nx = I1.shape
        u, v = _f64(u).copy(), _f64(v).copy()
        iters = (C.c_int * warps)()
        self._fn("hs_single_scale", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                 C.c_double, C.c_int, C.c_int, _ip)(_f64(I1), _f64(I2), u, v, nx, ny, alpha, warps, TOL, maxiter,
                                                    verbose, iters)
        return u, v, list(iters)

    def hs_pyramidal(self, I1, I2, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150,
                     verbose=0):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        iters = (C.c_int * (warps * nscales))()
        rc = self._fn("hs_pyramidal", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                      C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, _ip)(
            _f64(I1), _f64(I2), u, v, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter, verbose, iters)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v, np.array(list(iters)).reshape(nscales, warps)

    def brox_spatial(self, I1, I2, alpha=50.0, gamma=10.0, nscales=10, nu=0.5, TOL=1e-4, inner=1, outer=15,
                     verbose=0):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        iters = (C.c_int * (inner * outer * nscales))()
        rc = self._fn("brox_spatial", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _ip)(
            _f64(I1), _f64(I2), u, v, nx, ny, alpha, gamma, nscales, nu, TOL, inner, outer, verbose, iters)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v, np.array(list(iters)).reshape(nscales, inner * outer)

    def robust_expo(self, I1, I2, method=1, alpha=50.0, gamma=10.0, lam=1.0, nscales=5, nu=0.5, TOL=1e-4, inner=1, outer=15,
                    verbose=0):
        """robust_expo_methods, one channel -> (u, v, sweep counts [scale][outer * inner])"""
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        iters = (C.c_int * (inner * outer * nscales))()
        rc = self._fn("robust_expo", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                      C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _ip)(
            _f64(I1), _f64(I2), u, v, nx, ny, method, alpha, gamma, lam, nscales, nu, TOL, inner, outer, verbose, iters)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v, np.array(list(iters)).reshape(nscales, inner * outer)

    def gaussian_dirichlet(self, I, sigma):
        out = _f64(I).copy()
        self._fn("gaussian_dirichlet", None, _dp, C.c_int, C.c_int, C.c_double)(out, I.shape[1], I.shape[0], sigma)
        return out

    def rexpo_exponential(self, Ix, Iy, alpha, lam, method):
        expo = np.empty(Ix.shape)
        self._fn("rexpo_exponential", None, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp)(
            _f64(Ix), _f64(Iy), Ix.size, alpha, lam, method, expo)
        return expo

    def brox_temporal(self, I, alpha=18.0, gamma=7.0, nscales=10, nu=0.75, TOL=1e-4, inner=1, outer=15, verbose=0):
        """I: (frames, ny, nx).  Returns u, v of shape (frames - 1, ny, nx) and the sweep counts [scale][solve]."""
        frames, ny, nx = I.shape
        u, v = np.zeros((frames - 1, ny, nx)), np.zeros((frames - 1, ny, nx))
        iters = (C.c_int * (inner * outer * nscales))()
        rc = self._fn("brox_temporal", C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _ip)(
            _f64(I), u, v, nx, ny, frames, alpha, gamma, nscales, nu, TOL, inner, outer, verbose, iters)
        if rc == 1:
            raise ValueError("GaussianSmooth: sigma too large")
        if rc:
            raise ValueError("The method needs more than two frames")
        return u, v, np.array(list(iters)).reshape(nscales, inner * outer)



class Ref(_Lib):
    """The compiled reference itself (oracle/_ref/libofref.so via oracle/ref_shim.cpp)."""
    prefix = "ref_"
    kind = "reference"

    def tvl1occ_multiscale(self, I_1, I0, I1, filtI0=None, lam=0.15, alpha=0.01, beta=0.15, theta=0.3, nscales=3, zfactor=0.5,
                           warps=2, epsilon=0.01, verbose=0):
        """the reference's Dual_TVL1_optic_flow_multiscale (tvl1occflow.h) on a zero-filled heap -> (u1, u2, chi)"""
        ny, nx = I0.shape
        filtI0 = I0 if filtI0 is None else filtI0
        u1, u2, chi = np.zeros((ny, nx)), np.zeros((ny, nx)), np.zeros((ny, nx))
        self._fn("tvl1occ_multiscale", None, *([_dp] * 7), C.c_int, C.c_int, *([C.c_double] * 4), C.c_int, C.c_double, C.c_int,
                 C.c_double, C.c_int)(_f64(I_1), _f64(I0), _f64(I1), _f64(filtI0), u1, u2, chi, nx, ny, lam, alpha, beta, theta,
                                      nscales, zfactor, warps, epsilon, verbose)
        return u1, u2, chi

    def occ_solver_u(self, v1, v2, chi, g, theta, beta, fresh=True):
        """One call of the reference's Solver_wrt_u -> (u1, u2).  fresh: zero dual planes (oracle/ref_shim.cpp); otherwise
        the ones the previous call left in the function's statics."""
        ny, nx = v1.shape
        u1, u2 = np.empty((ny, nx)), np.empty((ny, nx))
        self._fn("occ_solver_u", None, *([_dp] * 6), C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            u1, u2, _f64(v1), _f64(v2), _f64(chi), _f64(g), theta, beta, nx, ny, int(fresh))
        return u1, u2

    def occ_solver_chi(self, u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vf1, Vf2, Vb1, Vb2, g, lam, theta, alpha,
                       beta, tau_chi, tau_eta, fresh=True):
        """One call of the reference's Solver_wrt_chi (100 iterations).  fresh: start from a zero dual variable
        (oracle/ref_shim.cpp); otherwise continue with the one the previous call left in the function's statics."""
        ny, nx = u1.shape
        chi = _f64(chi).copy()
        self._fn("occ_solver_chi", None, *([_dp] * 14), *([C.c_double] * 6), C.c_int, C.c_int, C.c_int)(
            _f64(u1), _f64(u2), chi, _f64(I1wx), _f64(I1wy), _f64(I_1wx), _f64(I_1wy), _f64(rho1_c), _f64(rho3_c), _f64(Vf1),
            _f64(Vf2), _f64(Vb1), _f64(Vb2), _f64(g), lam, theta, alpha, beta, tau_chi, tau_eta, nx, ny, int(fresh))
        return chi

    def __init__(self):
        super().__init__(REF_SO)

    def set_num_threads(self, n):
        self._fn("set_num_threads", None, C.c_int)(n)

    def max_threads(self):
        return self._fn("max_threads", C.c_int)()

    def dxx(self, f): return self._second("Dxx", f)
    def dyy(self, f): return self._second("Dyy", f)
    def dxy(self, f): return self._second("Dxy", f)

    def tvl1_single_scale(self, I0, I1, u1, u2, tau=0.25, lam=0.15, theta=0.3, warps=5, epsilon=0.01,
                          verbose=0):
        ny, nx = I0.shape
        u1, u2 = _f64(u1).copy(), _f64(u2).copy()
        self._fn("tvl1_single_scale", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                 C.c_double, C.c_int, C.c_double, C.c_int)(
            _f64(I0).copy(), _f64(I1).copy(), u1, u2, nx, ny, tau, lam, theta, warps, epsilon, verbose)
        return u1, u2

    def tvl1_multiscale(self, I0, I1, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5,
                        epsilon=0.01, verbose=0):
        ny, nx = I0.shape
        u1, u2 = np.zeros((ny, nx)), np.zeros((ny, nx))
        rc = self._fn("tvl1_multiscale", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double,
                      C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int)(
            _f64(I0).copy(), _f64(I1).copy(), u1, u2, nx, ny, tau, lam, theta, nscales, zfactor, warps,
            epsilon, verbose)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u1, u2

    def hs_single_scale(self, I1, I2, u, v, alpha=7.0, warps=10, TOL=1e-4, maxiter=150, verbose=0):
        ny, nx = I1.shape
        u, v = _f64(u).copy(), _f64(v).copy()
        self._fn("hs_single_scale", None, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                 C.c_double, C.c_int, C.c_int)(_f64(I1), _f64(I2), u, v, nx, ny, alpha, warps, TOL, maxiter, verbose)
        return u, v

    def hs_pyramidal(self, I1, I2, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150,
                     verbose=0):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        rc = self._fn("hs_pyramidal", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                      C.c_double, C.c_int, C.c_double, C.c_int, C.c_int)(
            _f64(I1), _f64(I2), u, v, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter, verbose)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v

    def brox_spatial(self, I1, I2, alpha=50.0, gamma=10.0, nscales=10, nu=0.5, TOL=1e-4, inner=1, outer=15,
                     verbose=0):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        rc = self._fn("brox_spatial", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            _f64(I1), _f64(I2), u, v, nx, ny, alpha, gamma, nscales, nu, TOL, inner, outer, verbose)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v

    def robust_expo(self, I1, I2, method=1, alpha=50.0, gamma=10.0, lam=1.0, nscales=5, nu=0.5, TOL=1e-4, inner=1, outer=15,
                    verbose=0):
        """the reference's robust_expo_methods with nzz = 1 -> (u, v)"""
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        rc = self._fn("robust_expo", C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            _f64(I1), _f64(I2), u, v, nx, ny, 1, method, alpha, gamma, lam, nscales, nu, TOL, inner, outer, verbose)
        if rc:
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v

    def gaussian_bc(self, I, sigma, bc):
        out = _f64(I).copy()
        self._fn("gaussian_bc", None, _dp, C.c_int, C.c_int, C.c_double, C.c_int)(out, I.shape[1], I.shape[0], sigma, bc)
        return out

    def rexpo_exponential(self, Ix, Iy, alpha, lam, method):
        expo = np.empty(Ix.shape)
        self._fn("rexpo_exponential", None, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _dp)(
            _f64(Ix), _f64(Iy), Ix.size, Ix.size, 1, alpha, lam, method, expo)
        return expo

    def brox_temporal(self, I, alpha=18.0, gamma=7.0, nscales=10, nu=0.75, TOL=1e-4, inner=1, outer=15, verbose=0):
        frames, ny, nx = I.shape
        u, v = np.zeros((frames - 1, ny, nx)), np.zeros((frames - 1, ny, nx))
        rc = self._fn("brox_temporal", C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)(
            _f64(I), u, v, nx, ny, frames, alpha, gamma, nscales, nu, TOL, inner, outer, verbose)
        if rc == 1:
            raise ValueError("GaussianSmooth: sigma too large")
        if rc:
            raise ValueError("The method needs more than two frames")
        return u, v


class Ref32:
    """The reference's own FLOAT build (src/of.h with OFPIX_DOUBLE off), double planes converted at the boundary:
    what float storage costs the reference itself (oracle variant 2)."""
    kind = "reference-f32"

    def __init__(self):
        if not os.path.exists(REF32_SO):
            raise FileNotFoundError(REF32_SO + " (run `make -C oracle` where /root/reference is present)")
        self.lib = C.CDLL(REF32_SO)

    def set_num_threads(self, n):
        f = self.lib.ref32_set_num_threads
        f.restype, f.argtypes = None, [C.c_int]
        f(n)

    def tvl1_multiscale(self, I0, I1, tau=0.25, lam=0.15, theta=0.3, nscales=5, zfactor=0.5, warps=5, epsilon=0.01, verbose=0):
        ny, nx = I0.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        f = self.lib.ref32_tvl1_multiscale
        f.restype = C.c_int
        f.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int,
                      C.c_double, C.c_int]
        if f(_f64(I0), _f64(I1), u, v, nx, ny, tau, lam, theta, nscales, zfactor, warps, epsilon, verbose):
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v

    def hs_pyramidal(self, I1, I2, alpha=7.0, nscales=10, zfactor=0.5, warps=10, TOL=1e-4, maxiter=150, verbose=0):
        ny, nx = I1.shape
        u, v = np.zeros((ny, nx)), np.zeros((ny, nx))
        f = self.lib.ref32_hs_pyramidal
        f.restype = C.c_int
        f.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
        if f(_f64(I1), _f64(I2), u, v, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter, verbose):
            raise ValueError("GaussianSmooth: sigma too large")
        return u, v
