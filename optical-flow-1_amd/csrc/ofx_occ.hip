// ofx_occ.hip -- operators next to the hot path: the colour variants of warp and normalisation
// (src/bicubic_interpolation.cpp:381-405, src/utils.cpp:333-501) and the deterministic building blocks of TV-L1 with
// occlusions (me_median_filtering src/utils.cpp:150-213, Solver_wrt_v / Solver_wrt_chi
// src/tvl1occflow_solvers.cpp:56-147,218-337).  Host double planes in / out; arrays are kept as doubles on the device
// whatever the context's storage precision (these are operator-level entry points, not part of a device-resident
// solve).  Every result is bit-identical to the reference's (tests/test_gpu_occ.py against the oracle, which is pinned
// against the compiled reference in tests/test_oracle_vs_ref.py).
#include "ofx_ops.h"
#include "ofx_device.h"

#define OCC_IS_ZERO 1E-10      // src/tvl1occflow_constants.h:31
#define OCC_THR_CHI 0.75       // src/tvl1occflow_constants.h:32
#define OCC_MM_BLOCKS 256
#define OFX_MEDIAN_MAX_W 9     // largest window side of ofx_me_median_filtering

static inline int occ_grid1d(size_t n) { return (int) ((n + 255) / 256); }

namespace {

struct Dev {                   // upload / download helpers on the context's arena and stream
    ofx_ctx *ctx;
    int in(const double *h, double **d, size_t n)
    {
        OFX_TRY(ofx_alloc(ctx, n, d));
        OFX_HIP(ctx, hipMemcpyAsync(*d, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        return OFX_OK;
    }
    int out(const double *d, double *h, size_t n)
    {
        OFX_HIP(ctx, hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        return OFX_OK;
    }
    int sync()
    {
        OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return OFX_OK;
    }
};

}   // namespace

// ---- bicubic_interpolation_warp_color: one thread per (pixel, channel) -----------------------------------------------
__global__ void k_warp_color(const double *__restrict__ in, const double *__restrict__ u, const double *__restrict__ v,
                             double *__restrict__ out, int nx, int ny, int nz, int border_out)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) nx * ny * nz) return;
    const int k = (int) (e % nz);
    const size_t p = e / nz;
    const int i = (int) (p / nx), j = (int) (p % nx);
    const BicubicTaps t = bicubic_taps(j + u[p], i + v[p], nx, ny);
    double r = 0.0;
    if (!(t.out && border_out)) {
        double c[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double v0 = in[((size_t) t.row[0] * nx + t.col[q]) * nz + k], v1 = in[((size_t) t.row[1] * nx + t.col[q]) * nz + k];
            const double v2 = in[((size_t) t.row[2] * nx + t.col[q]) * nz + k], v3 = in[((size_t) t.row[3] * nx + t.col[q]) * nz + k];
            c[q] = cubic_cell(v0, v1, v2, v3, t.fy);
        }
        r = cubic_cell(c[0], c[1], c[2], c[3], t.fx);
    }
    out[e] = r;
}

extern "C" int ofx_bicubic_warp_color(ofx_ctx *ctx, const double *input, const double *u, const double *v, double *output,
                                      int nx, int ny, int nz, int border_out)
{
    OFX_ENTER(ctx);
    if (!input || !u || !v || !output) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    if (nx < 1 || ny < 1 || nz < 1 || (long long) nx * ny * nz > 0x7fffffffLL)
        return ofx_fail(ctx, OFX_ERR_ARG, "bicubic_warp_color: bad size %dx%dx%d", nx, ny, nz);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *di, *du, *dv, *dout;
    OFX_TRY(d.in(input, &di, n * nz));
    OFX_TRY(d.in(u, &du, n));
    OFX_TRY(d.in(v, &dv, n));
    OFX_TRY(ofx_alloc(ctx, n * nz, &dout));
    hipLaunchKernelGGL(k_warp_color, dim3(occ_grid1d(n * nz)), dim3(256), 0, ctx->stream, di, du, dv, dout, nx, ny, nz, border_out);
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(d.out(dout, output, n * nz));
    return d.sync();
}

// ---- min / max of strided element sets (exact: order-independent) -------------------------------------------------------
// part[b] / part[OCC_MM_BLOCKS + b] = min / max over elements first + t * stride (t = 0 .. count - 1) of up to two arrays
__global__ void k_mm_partial(const double *__restrict__ a, const double *__restrict__ b, size_t first, size_t stride,
                             size_t count, double *__restrict__ part)
{
    double lo = a[first], hi = lo;
    for (size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (size_t) gridDim.x * blockDim.x) {
        const double x = a[first + t * stride];
        lo = x < lo ? x : lo; hi = x > hi ? x : hi;
        if (b) {
            const double y = b[first + t * stride];
            lo = y < lo ? y : lo; hi = y > hi ? y : hi;
        }
    }
    lo = wave_allreduce_min(lo);
    hi = wave_allreduce_max(hi);
    __shared__ double slo[4], shi[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[w] = lo; shi[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { lo = slo[k] < lo ? slo[k] : lo; hi = shi[k] > hi ? shi[k] : hi; }
        part[blockIdx.x] = lo;
        part[OCC_MM_BLOCKS + blockIdx.x] = hi;
    }
}
// mm[0] = min, mm[1] = max over the partials; `acc` != 0 merges with what mm already holds
__global__ void k_mm_final(const double *__restrict__ part, int nblocks, double *__restrict__ mm, int acc)
{
    if (threadIdx.x != 0) return;
    double lo = part[0], hi = part[OCC_MM_BLOCKS];
    for (int i = 1; i < nblocks; i++) {
        lo = part[i] < lo ? part[i] : lo;
        hi = part[OCC_MM_BLOCKS + i] > hi ? part[OCC_MM_BLOCKS + i] : hi;
    }
    if (acc) { lo = mm[0] < lo ? mm[0] : lo; hi = mm[1] > hi ? mm[1] : hi; }
    mm[0] = lo;
    mm[1] = hi;
}
static int occ_minmax(ofx_ctx *ctx, const double *a, const double *b, size_t first, size_t stride, size_t count, double *part,
                      double *mm, int acc)
{
    int nb = occ_grid1d(count);
    nb = nb < 1 ? 1 : (nb > OCC_MM_BLOCKS ? OCC_MM_BLOCKS : nb);
    hipLaunchKernelGGL(k_mm_partial, dim3(nb), dim3(256), 0, ctx->stream, a, b, first, stride, count, part);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_mm_final, dim3(1), dim3(64), 0, ctx->stream, (const double *) part, nb, mm, acc);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// 255 (x - min) / den per element, or a copy when `guard` and den <= 0 (image_normalization_2_color / _4); channel c of
// an nz-interleaved array uses mm[2 c], mm[2 c + 1]
__global__ void k_norm_map(const double *__restrict__ in, double *__restrict__ out, size_t n, int nz,
                           const double *__restrict__ mm, int guard)
{
    const size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int c = (int) (e % nz);
    const double lo = mm[2 * c], den = mm[2 * c + 1] - lo;
    const double x = in[e];
    out[e] = (!guard || den > 0) ? 255.0 * (x - lo) / den : x;
}

extern "C" int ofx_image_normalization_2_color(ofx_ctx *ctx, const double *I1, const double *I2, double *I1n, double *I2n,
                                               int size, int nz)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !I1n || !I2n) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    // the reference scans elements c + nz, c + 2 nz, ... for channel c: a size that is not a multiple of nz reads past
    // the arrays there
    if (nz < 1 || size < nz || size % nz) return ofx_fail(ctx, OFX_ERR_ARG, "image_normalization_2_color: size %d, nz %d", size, nz);
    Dev d{ctx};
    double *a, *b, *oa, *ob, *part, *mm;
    OFX_TRY(d.in(I1, &a, size));
    OFX_TRY(d.in(I2, &b, size));
    OFX_TRY(ofx_alloc(ctx, (size_t) size, &oa));
    OFX_TRY(ofx_alloc(ctx, (size_t) size, &ob));
    OFX_TRY(ofx_alloc(ctx, (size_t) 2 * OCC_MM_BLOCKS, &part));
    OFX_TRY(ofx_alloc(ctx, (size_t) 2 * nz, &mm));
    for (int c = 0; c < nz; c++) OFX_TRY(occ_minmax(ctx, a, b, (size_t) c, (size_t) nz, (size_t) size / nz, part, mm + 2 * c, 0));
    hipLaunchKernelGGL(k_norm_map, dim3(occ_grid1d(size)), dim3(256), 0, ctx->stream, (const double *) a, oa, (size_t) size, nz,
                       (const double *) mm, 1);
    hipLaunchKernelGGL(k_norm_map, dim3(occ_grid1d(size)), dim3(256), 0, ctx->stream, (const double *) b, ob, (size_t) size, nz,
                       (const double *) mm, 1);
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(d.out(oa, I1n, size));
    OFX_TRY(d.out(ob, I2n, size));
    return d.sync();
}

// joint min / max of k images, then the map; guard = 1: copy when den <= 0 (_4), 0: always divide (_3)
static int occ_normalize_joint(ofx_ctx *ctx, int k, const double *const *in, double *const *out, int size, int guard)
{
    Dev d{ctx};
    double *dv[4], *ov[4], *part, *mm;
    for (int q = 0; q < k; q++) {
        OFX_TRY(d.in(in[q], &dv[q], size));
        OFX_TRY(ofx_alloc(ctx, (size_t) size, &ov[q]));
    }
    OFX_TRY(ofx_alloc(ctx, (size_t) 2 * OCC_MM_BLOCKS, &part));
    OFX_TRY(ofx_alloc(ctx, (size_t) 2, &mm));
    for (int q = 0; q < k; q++) OFX_TRY(occ_minmax(ctx, dv[q], nullptr, 0, 1, (size_t) size, part, mm, q > 0));
    for (int q = 0; q < k; q++)
        hipLaunchKernelGGL(k_norm_map, dim3(occ_grid1d(size)), dim3(256), 0, ctx->stream, (const double *) dv[q], ov[q],
                           (size_t) size, 1, (const double *) mm, guard);
    OFX_LAUNCH_CHECK(ctx);
    for (int q = 0; q < k; q++) OFX_TRY(d.out(ov[q], out[q], size));
    return d.sync();
}

extern "C" int ofx_image_normalization_3(ofx_ctx *ctx, double *I0, double *I1, double *I2, int size)
{
    OFX_ENTER(ctx);
    if (!I0 || !I1 || !I2 || size < 1) return ofx_fail(ctx, OFX_ERR_ARG, "image_normalization_3: NULL pointer / size");
    const double *in[3] = {I0, I1, I2};
    double *out[3] = {I0, I1, I2};
    return occ_normalize_joint(ctx, 3, in, out, size, 0);        // the reference has no den > 0 test here (:412-450)
}

extern "C" int ofx_image_normalization_4(ofx_ctx *ctx, const double *I_1, const double *I0, const double *I1,
                                         const double *filtI0, double *I_1n, double *I0n, double *I1n, double *filtI0n,
                                         int size)
{
    OFX_ENTER(ctx);
    if (!I_1 || !I0 || !I1 || !filtI0 || !I_1n || !I0n || !I1n || !filtI0n || size < 1)
        return ofx_fail(ctx, OFX_ERR_ARG, "image_normalization_4: NULL pointer / size");
    const double *in[4] = {I_1, I0, I1, filtI0};
    double *out[4] = {I_1n, I0n, I1n, filtI0n};
    return occ_normalize_joint(ctx, 4, in, out, size, 1);
}

// ---- me_median_filtering: element [count / 2] of the sorted w x w window, mirrored indices (-1 -> 0, n -> n - 1) ----------
__global__ void k_median(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int w)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= nx || y >= ny) return;
    const int border = w >> 1;
    double win[OFX_MEDIAN_MAX_W * OFX_MEDIAN_MAX_W];
    int n = 0;
    for (int yy = y - border; yy <= y + border; yy++)
        for (int xx = x - border; xx <= x + border; xx++) {
            int x0 = xx, y0 = yy;
            if (x0 < 0) x0 = -x0 - 1;
            if (x0 >= nx) x0 = 2 * nx - x0 - 1;
            if (y0 < 0) y0 = -y0 - 1;
            if (y0 >= ny) y0 = 2 * ny - y0 - 1;
            win[n++] = in[(size_t) y0 * nx + x0];
        }
    // partial selection sort up to position n / 2 (which of several equal values ends up there cannot matter)
    const int m = n / 2;
    for (int a = 0; a <= m; a++) {
        int best = a;
        for (int b = a + 1; b < n; b++)
            if (win[b] < win[best]) best = b;
        const double t = win[a];
        win[a] = win[best];
        win[best] = t;
    }
    out[(size_t) y * nx + x] = win[m];
}

extern "C" int ofx_me_median_filtering(ofx_ctx *ctx, double *in, int nx, int ny, int wsize)
{
    OFX_ENTER(ctx);
    if (!in) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    if (nx < 1 || ny < 1 || (long long) nx * ny > 0x7fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "median: bad size %dx%d", nx, ny);
    // the mirrored index -x - 1 / 2 n - x - 1 must fall inside the image, as it has to in the reference
    if (wsize < 1 || wsize > OFX_MEDIAN_MAX_W || (wsize >> 1) > nx || (wsize >> 1) > ny)
        return ofx_fail(ctx, OFX_ERR_ARG, "median: window %d (1..%d, half window within the image)", wsize, OFX_MEDIAN_MAX_W);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *di, *dout;
    OFX_TRY(d.in(in, &di, n));
    OFX_TRY(ofx_alloc(ctx, n, &dout));
    hipLaunchKernelGGL(k_median, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)), dim3(64, 4), 0, ctx->stream, (const double *) di, dout, nx,
                       ny, wsize);
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(d.out(dout, in, n));
    return d.sync();
}

// ---- Solver_wrt_v (src/tvl1occflow_solvers.cpp:56-147): pointwise ---------------------------------------------------------
struct OccV {
    const double *u1, *u2, *chi, *I1wx, *I1wy, *I_1wx, *I_1wy, *rho1_c, *rho3_c, *grad1, *grad3;
    double *v1, *v2, *Vfwd_1, *Vfwd_2, *Vbck_1, *Vbck_2;
};
__global__ void k_occ_v(OccV a, int size, double alpha, double theta, double lambda)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    const double l_t = lambda * theta;
    const double _1pat = 1. + alpha * theta;
    const double at_d_1pat = alpha * theta / _1pat;
    const double lt_d_1pat = 2. * lambda * theta / _1pat;
    const double u1 = a.u1[i], u2 = a.u2[i];
    double d1, d2;
    const double ix = a.I1wx[i], iy = a.I1wy[i], g1 = a.grad1[i];
    const double rho1 = a.rho1_c[i] + (ix * u1 + iy * u2);
    if (rho1 < -l_t * g1) { d1 = l_t * ix; d2 = l_t * iy; }
    else if (rho1 > l_t * g1) { d1 = -l_t * ix; d2 = -l_t * iy; }
    else if (g1 < OCC_IS_ZERO) { d1 = 0; d2 = 0; }
    else { d1 = -rho1 * ix / g1; d2 = -rho1 * iy / g1; }
    const double f1 = u1 + d1, f2 = u2 + d2;
    const double jx = a.I_1wx[i], jy = a.I_1wy[i], g3 = a.grad3[i];
    const double rho3 = a.rho3_c[i] - (jx * u1 + jy * u2);
    const double A = rho3 + at_d_1pat * (jx * u1 + jy * u2);
    double b1, b2;
    if (A < -lt_d_1pat * g3) {
        d1 = -lt_d_1pat * jx; d2 = -lt_d_1pat * jy;
        b1 = (u1 / _1pat) + d1; b2 = (u2 / _1pat) + d2;
    } else if (A > lt_d_1pat * g3) {
        d1 = lt_d_1pat * jx; d2 = lt_d_1pat * jy;
        b1 = (u1 / _1pat) + d1; b2 = (u2 / _1pat) + d2;
    } else {
        if (g3 < OCC_IS_ZERO) { d1 = 0; d2 = 0; }
        else { d1 = rho3 * jx / g3; d2 = rho3 * jy / g3; }
        b1 = u1 + d1; b2 = u2 + d2;
    }
    a.Vfwd_1[i] = f1; a.Vfwd_2[i] = f2;
    a.Vbck_1[i] = b1; a.Vbck_2[i] = b2;
    const bool fwd = a.chi[i] < OCC_THR_CHI;
    a.v1[i] = fwd ? f1 : b1;
    a.v2[i] = fwd ? f2 : b2;
}

extern "C" int ofx_solver_wrt_v(ofx_ctx *ctx, const double *u1, const double *u2, double *v1, double *v2, const double *chi,
                                const double *I1wx, const double *I1wy, const double *I_1wx, const double *I_1wy,
                                const double *rho1_c, const double *rho3_c, double *Vfwd_1, double *Vfwd_2, double *Vbck_1,
                                double *Vbck_2, const double *grad1, const double *grad3, double alpha, double theta,
                                double lambda, int nx, int ny)
{
    OFX_ENTER(ctx);
    const double *ins[11] = {u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, grad1, grad3};
    double *outs[6] = {v1, v2, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2};
    for (auto p : ins) if (!p) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_v: NULL pointer");
    for (auto p : outs) if (!p) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_v: NULL pointer");
    if (nx < 1 || ny < 1 || (long long) nx * ny > 0x7fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_v: bad size %dx%d", nx, ny);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *di[11], *dout[6];
    for (int k = 0; k < 11; k++) OFX_TRY(d.in(ins[k], &di[k], n));
    for (int k = 0; k < 6; k++) OFX_TRY(ofx_alloc(ctx, n, &dout[k]));
    const OccV a = {di[0], di[1], di[2], di[3], di[4], di[5], di[6], di[7], di[8], di[9], di[10],
                    dout[0], dout[1], dout[2], dout[3], dout[4], dout[5]};
    hipLaunchKernelGGL(k_occ_v, dim3(occ_grid1d(n)), dim3(256), 0, ctx->stream, a, (int) n, alpha, theta, lambda);
    OFX_LAUNCH_CHECK(ctx);
    for (int k = 0; k < 6; k++) OFX_TRY(d.out(dout[k], outs[k], n));
    return d.sync();
}

// ---- Solver_wrt_chi (src/tvl1occflow_solvers.cpp:218-337), dual variable eta as explicit state -----------------------------------
// iteration = two launches: (a) eta += tau_eta g grad(chi), projected onto the unit ball (:33-53) -- reads chi of the
// right / lower neighbour; (b) chi += tau_chi (div(g eta) - F - G - beta div u), clamped to [0, 1] -- reads eta of the left
// / upper neighbour.  The kernel boundary between them is the only synchronisation needed.
__global__ void k_occ_eta(const double *__restrict__ chi, const double *__restrict__ g, double *__restrict__ eta1,
                          double *__restrict__ eta2, int nx, int ny, double tau_eta)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    const double c = chi[p];
    const double chix = (j < nx - 1) ? chi[p + 1] - c : 0.0;          // forward_gradient, src/operators.cpp:86-125
    const double chiy = (i < ny - 1) ? chi[p + nx] - c : 0.0;
    double e1 = eta1[p] + tau_eta * g[p] * chix;
    double e2 = eta2[p] + tau_eta * g[p] * chiy;
    const double norm2 = e1 * e1 + e2 * e2;
    if (norm2 < OCC_IS_ZERO) {
        e1 = 0.0;
        e2 = 0.0;
    } else {
        const double norm = sqrt(norm2);
        e1 = e1 / norm;
        e2 = e2 / norm;
    }
    eta1[p] = e1;
    eta2[p] = e2;
}

__global__ void k_occ_divu(const double *__restrict__ u1, const double *__restrict__ u2, double *__restrict__ div, int nx, int ny)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    const double al = j > 0 ? u1[p - 1] : 0.0, bu = i > 0 ? u2[p - nx] : 0.0;
    div[p] = div_backward(u1[p], al, u2[p], bu, j == 0, j == nx - 1, i == 0, i == ny - 1);
}

struct OccChi {
    const double *u1, *u2, *I1wx, *I1wy, *I_1wx, *I_1wy, *rho1_c, *rho3_c, *Vf1, *Vf2, *Vb1, *Vb2, *g, *eta1, *eta2, *div_u;
    double *chi;
};
__global__ void k_occ_chi(OccChi a, int nx, int ny, double lambda, double theta, double alpha, double beta, double tau_chi)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    // divergence of (g eta1, g eta2), src/operators.cpp:35-78
    const double ac = a.g[p] * a.eta1[p], bc = a.g[p] * a.eta2[p];
    const double al = j > 0 ? a.g[p - 1] * a.eta1[p - 1] : 0.0, bu = i > 0 ? a.g[p - nx] * a.eta2[p - nx] : 0.0;
    const double div_eta = div_backward(ac, al, bc, bu, j == 0, j == nx - 1, i == 0, i == ny - 1);
    const double u1 = a.u1[p], u2 = a.u2[p];
    const double f1 = a.Vf1[p], f2 = a.Vf2[p], b1 = a.Vb1[p], b2 = a.Vb2[p];
    const double rho1 = a.rho1_c[p] + (a.I1wx[p] * f1 + a.I1wy[p] * f2);
    const double abs_rho1 = (rho1 < 0.) ? -rho1 : rho1;
    const double rho3 = a.rho3_c[p] - (a.I_1wx[p] * b1 + a.I_1wy[p] * b2);
    const double abs_rho3 = (rho3 < 0.) ? -rho3 : rho3;
    double c = a.chi[p];
    double F, G;
    if (c < 0.5) {
        F = -lambda * abs_rho1;
        G = -(0.5 / theta) * ((f1 - u1) * (f1 - u1) + (f2 - u2) * (f2 - u2));
    } else {
        F = lambda * abs_rho3;
        G = (0.5 / theta) * ((b1 - u1) * (b1 - u1) + (b2 - u2) * (b2 - u2)) + alpha * theta * (b1 * b1 + b2 * b2);
    }
    c = c + tau_chi * (div_eta - F - G - beta * a.div_u[p]);
    if (c > 1.) c = 1.;
    else if (c < 0.) c = 0.;
    a.chi[p] = c;
}

extern "C" int ofx_solver_wrt_chi(ofx_ctx *ctx, const double *u1, const double *u2, double *chi, const double *I1wx,
                                  const double *I1wy, const double *I_1wx, const double *I_1wy, const double *rho1_c,
                                  const double *rho3_c, const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1,
                                  const double *Vbck_2, const double *g, double lambda, double theta, double alpha, double beta,
                                  double tau_chi, double tau_eta, int nx, int ny, double *eta1, double *eta2, int n_iter)
{
    OFX_ENTER(ctx);
    const double *ins[16] = {u1, u2, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2, g, chi, eta1, eta2};
    for (auto p : ins) if (!p) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_chi: NULL pointer");
    if (nx < 2 || ny < 2 || (long long) nx * ny > 0x7fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_chi: bad size %dx%d", nx, ny);
    if (n_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_chi: n_iter=%d", n_iter);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *di[16], *div_u;
    for (int k = 0; k < 16; k++) OFX_TRY(d.in(ins[k], &di[k], n));
    OFX_TRY(ofx_alloc(ctx, n, &div_u));
    const dim3 grid(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)), block(64, 4);
    hipLaunchKernelGGL(k_occ_divu, grid, block, 0, ctx->stream, (const double *) di[0], (const double *) di[1], div_u, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    const OccChi a = {di[0], di[1], di[2], di[3], di[4], di[5], di[6], di[7], di[8], di[9], di[10], di[11], di[12], di[14], di[15],
                      div_u, di[13]};
    for (int it = 0; it < n_iter; it++) {
        hipLaunchKernelGGL(k_occ_eta, grid, block, 0, ctx->stream, (const double *) di[13], (const double *) di[12], di[14], di[15],
                           nx, ny, tau_eta);
        hipLaunchKernelGGL(k_occ_chi, grid, block, 0, ctx->stream, a, nx, ny, lambda, theta, alpha, beta, tau_chi);
    }
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(d.out(di[13], chi, n));
    OFX_TRY(d.out(di[14], eta1, n));
    OFX_TRY(d.out(di[15], eta2, n));
    return d.sync();
}
