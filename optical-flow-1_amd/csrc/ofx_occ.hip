// ofx_occ.hip -- operators next to the hot path: the colour variants of warp and normalisation
// (src/bicubic_interpolation.cpp:381-405, src/utils.cpp:333-501) and the deterministic building blocks of TV-L1 with
// occlusions (me_median_filtering src/utils.cpp:150-213, Solver_wrt_v / Solver_wrt_chi
// src/tvl1occflow_solvers.cpp:56-147,218-337).  Host double planes in / out; arrays are kept as doubles on the device
// whatever the context's storage precision (these are operator-level entry points, not part of a device-resident
// solve).  Every result is bit-identical to the reference's (tests/test_gpu_occ.py against the oracle, which is pinned
// against the compiled reference in tests/test_oracle_vs_ref.py).
#include "ofx_ops.h"
#include "ofx_device.h"

#include <atomic>
#include <thread>

#define OCC_IS_ZERO 1E-10      // src/tvl1occflow_constants.h:31
#define OCC_THR_CHI 0.75       // src/tvl1occflow_constants.h:32
#define OCC_MM_BLOCKS 256
#define OFX_MEDIAN_MAX_W 9     // largest window side of ofx_me_median_filtering

static inline int occ_grid1d(size_t n) { return (int) ((n + 255) / 256); }

// Lockstep groups of TV-L1-with-occlusions solves (ofx_tvl1occ_batch): triple g = blockIdx.z works on planes offset by
// g * stride elements; bit g of mask = that triple is still iterating (its workgroups return at once otherwise, its state
// stays frozen).  The operator-level entry points pass {0, 1} with gridDim.z = 1.
struct OccGrp {
    size_t stride;
    unsigned mask;
};
#define OCC_ONE OccGrp{0, 1u}
static __device__ __forceinline__ bool occ_grp(const OccGrp &G, size_t &off)
{
    if (!((G.mask >> blockIdx.z) & 1u)) return false;
    off = (size_t) blockIdx.z * G.stride;
    return true;
}

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
__global__ void k_occ_v(OccV a, int size, double alpha, double theta, double lambda, OccGrp G)
{
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    size_t o;
    if (i0 >= size || !occ_grp(G, o)) return;
    const size_t i = o + i0;
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
    hipLaunchKernelGGL(k_occ_v, dim3(occ_grid1d(n)), dim3(256), 0, ctx->stream, a, (int) n, alpha, theta, lambda, OCC_ONE);
    OFX_LAUNCH_CHECK(ctx);
    for (int k = 0; k < 6; k++) OFX_TRY(d.out(dout[k], outs[k], n));
    return d.sync();
}

// ---- Solver_wrt_chi (src/tvl1occflow_solvers.cpp:218-337), dual variable eta as explicit state -----------------------------------
// iteration = two launches: (a) eta += tau_eta g grad(chi), projected onto the unit ball (:33-53) -- reads chi of the
// right / lower neighbour; (b) chi += tau_chi (div(g eta) - F - G - beta div u), clamped to [0, 1] -- reads eta of the left
// / upper neighbour.  The kernel boundary between them is the only synchronisation needed.
__global__ void k_occ_eta(const double *__restrict__ chi, const double *__restrict__ g, double *__restrict__ eta1,
                          double *__restrict__ eta2, int nx, int ny, double tau_eta, OccGrp G)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(G, o)) return;
    const size_t p = o + (size_t) i * nx + j;
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

__global__ void k_occ_divu(const double *__restrict__ u1, const double *__restrict__ u2, double *__restrict__ div, int nx, int ny,
                           OccGrp G)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(G, o)) return;
    const size_t p = o + (size_t) i * nx + j;
    const double al = j > 0 ? u1[p - 1] : 0.0, bu = i > 0 ? u2[p - nx] : 0.0;
    div[p] = div_backward(u1[p], al, u2[p], bu, j == 0, j == nx - 1, i == 0, i == ny - 1);
}

struct OccChi {
    const double *u1, *u2, *I1wx, *I1wy, *I_1wx, *I_1wy, *rho1_c, *rho3_c, *Vf1, *Vf2, *Vb1, *Vb2, *g, *eta1, *eta2, *div_u;
    double *chi;
};
__global__ void k_occ_chi(OccChi a, int nx, int ny, double lambda, double theta, double alpha, double beta, double tau_chi,
                          OccGrp grp)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(grp, o)) return;
    const size_t p = o + (size_t) i * nx + j;
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

// Several iterations per launch.  One iteration moves information by one pixel (eta reads chi of the right / lower
// neighbour, chi reads g eta of the left / upper one), so a workgroup that holds a CHI_T x CHI_T tile of (chi, g eta1, g eta2) in
// LDS can run n <= CHI_N iterations on its own: whatever enters from beyond the tile edge is wrong by then only within
// CHI_N pixels of the edge, and the (CHI_T - 2 CHI_N)^2 pixels in the middle are exactly what n global iterations give
// (same expressions, same order; pixels outside the image are never read -- the one-sided differences at the image border
// do not look there).  The state goes from one set of planes to another (tiles overlap), the per-pixel constants -- g, the
// two branches of F and G, beta div u -- are computed once per launch and stay in registers.  100 iterations = 20 launches
// instead of 200; a small level is a chain of launch latencies, so that is most of its time.
#define CHI_T 32
#ifndef CHI_N
#define CHI_N 5
#endif
#define CHI_IN (CHI_T - 2 * CHI_N)
struct OccChiState {
    const double *chi, *eta1, *eta2;     // in
    double *chi_o, *eta1_o, *eta2_o;     // out
};
__global__ __launch_bounds__(256) void k_occ_chi_fused(OccChi a, OccChiState st, int nx, int ny, int n_it, double lambda, double theta,
                                                       double alpha, double beta, double tau_chi, double tau_eta, OccGrp grp)
{
    __shared__ double s_chi[CHI_T][CHI_T + 1], s_g1[CHI_T][CHI_T + 1], s_g2[CHI_T][CHI_T + 1];
    size_t o;
    if (!occ_grp(grp, o)) return;
    const int tx = threadIdx.x, ty = threadIdx.y;                       // 32 x 8: four rows per thread
    const int j = (int) blockIdx.x * CHI_IN - CHI_N + tx;
    const int i0 = (int) blockIdx.y * CHI_IN - CHI_N + ty;
    constexpr int M = CHI_T / 8;
    double g[M], F0[M], G0[M], F1[M], G1[M], bd[M], c[M], e1[M], e2[M];
    bool in[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = i0 + 8 * m, ly = ty + 8 * m;
        in[m] = j >= 0 && j < nx && i >= 0 && i < ny;
        g[m] = F0[m] = G0[m] = F1[m] = G1[m] = bd[m] = c[m] = e1[m] = e2[m] = 0.0;
        if (in[m]) {
            const size_t p = o + (size_t) i * nx + j;
            g[m] = a.g[p];
            const double u1 = a.u1[p], u2 = a.u2[p];
            const double f1 = a.Vf1[p], f2 = a.Vf2[p], b1 = a.Vb1[p], b2 = a.Vb2[p];
            const double rho1 = a.rho1_c[p] + (a.I1wx[p] * f1 + a.I1wy[p] * f2);
            const double abs_rho1 = (rho1 < 0.) ? -rho1 : rho1;
            const double rho3 = a.rho3_c[p] - (a.I_1wx[p] * b1 + a.I_1wy[p] * b2);
            const double abs_rho3 = (rho3 < 0.) ? -rho3 : rho3;
            F0[m] = -lambda * abs_rho1;
            G0[m] = -(0.5 / theta) * ((f1 - u1) * (f1 - u1) + (f2 - u2) * (f2 - u2));
            F1[m] = lambda * abs_rho3;
            G1[m] = (0.5 / theta) * ((b1 - u1) * (b1 - u1) + (b2 - u2) * (b2 - u2)) + alpha * theta * (b1 * b1 + b2 * b2);
            bd[m] = beta * a.div_u[p];
            c[m] = st.chi[p];
            e1[m] = st.eta1[p];
            e2[m] = st.eta2[p];
        }
        s_chi[ly][tx] = c[m];
        s_g1[ly][tx] = g[m] * e1[m];
        s_g2[ly][tx] = g[m] * e2[m];
    }
    __syncthreads();
    const int txr = tx < CHI_T - 1 ? tx + 1 : tx, txl = tx > 0 ? tx - 1 : 0;        // beyond the tile: any value (see above)
    for (int it = 0; it < n_it; it++) {
#pragma unroll
        for (int m = 0; m < M; m++) {                                    // eta += tau_eta g grad(chi), projected (k_occ_eta)
            const int i = i0 + 8 * m, ly = ty + 8 * m, lyd = ly < CHI_T - 1 ? ly + 1 : ly;
            if (in[m]) {
                const double chix = (j < nx - 1) ? s_chi[ly][txr] - c[m] : 0.0;
                const double chiy = (i < ny - 1) ? s_chi[lyd][tx] - c[m] : 0.0;
                double a1 = e1[m] + tau_eta * g[m] * chix;
                double a2 = e2[m] + tau_eta * g[m] * chiy;
                const double norm2 = a1 * a1 + a2 * a2;
                if (norm2 < OCC_IS_ZERO) {
                    a1 = 0.0;
                    a2 = 0.0;
                } else {
                    const double norm = sqrt(norm2);
                    a1 = a1 / norm;
                    a2 = a2 / norm;
                }
                e1[m] = a1;
                e2[m] = a2;
                s_g1[ly][tx] = g[m] * a1;
                s_g2[ly][tx] = g[m] * a2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; m++) {                                    // chi += tau_chi (div(g eta) - F - G - beta div u), clamped (k_occ_chi)
            const int i = i0 + 8 * m, ly = ty + 8 * m, lyu = ly > 0 ? ly - 1 : 0;
            if (in[m]) {
                const double ac = g[m] * e1[m], bc = g[m] * e2[m];
                const double al = j > 0 ? s_g1[ly][txl] : 0.0, bu = i > 0 ? s_g2[lyu][tx] : 0.0;
                const double div_eta = div_backward(ac, al, bc, bu, j == 0, j == nx - 1, i == 0, i == ny - 1);
                double cc = c[m];
                const double F = cc < 0.5 ? F0[m] : F1[m], G = cc < 0.5 ? G0[m] : G1[m];
                cc = cc + tau_chi * (div_eta - F - G - bd[m]);
                if (cc > 1.) cc = 1.;
                else if (cc < 0.) cc = 0.;
                c[m] = cc;
                s_chi[ly][tx] = cc;
            }
        }
        __syncthreads();
    }
    if (tx >= CHI_N && tx < CHI_T - CHI_N) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = i0 + 8 * m, ly = ty + 8 * m;
            if (in[m] && ly >= CHI_N && ly < CHI_T - CHI_N) {
                const size_t p = o + (size_t) i * nx + j;
                st.chi_o[p] = c[m];
                st.eta1_o[p] = e1[m];
                st.eta2_o[p] = e2[m];
            }
        }
    }
}
// n_iter iterations of the chi solver on G sets of planes: (chi, eta1, eta2) in place, alt = three scratch planes per set
static int occ_chi_iterations(ofx_ctx *ctx, const OccChi &a, double *chi, double *eta1, double *eta2, double *alt, int nx, int ny,
                              int n_iter, double lambda, double theta, double alpha, double beta, double tau_chi, double tau_eta,
                              int G, const OccGrp &grp)
{
    const size_t n = (size_t) nx * ny * G;
    if (!ctx->chi_fuse) {
        const dim3 grid(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G), block(64, 4);
        for (int k = 0; k < n_iter; k++) {
            hipLaunchKernelGGL(k_occ_eta, grid, block, 0, ctx->stream, (const double *) chi, a.g, eta1, eta2, nx, ny, tau_eta, grp);
            hipLaunchKernelGGL(k_occ_chi, grid, block, 0, ctx->stream, a, nx, ny, lambda, theta, alpha, beta, tau_chi, grp);
        }
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    }
    double *cur[3] = {chi, eta1, eta2}, *nxt[3] = {alt, alt + n, alt + 2 * n};
    const dim3 grid(ofx_cdiv(nx, CHI_IN), ofx_cdiv(ny, CHI_IN), G), block(CHI_T, 8);
    int launches = 0;
    for (int k = 0; k < n_iter; k += CHI_N, launches++) {
        const OccChiState st = {cur[0], cur[1], cur[2], nxt[0], nxt[1], nxt[2]};
        hipLaunchKernelGGL(k_occ_chi_fused, grid, block, 0, ctx->stream, a, st, nx, ny, n_iter - k < CHI_N ? n_iter - k : CHI_N, lambda,
                           theta, alpha, beta, tau_chi, tau_eta, grp);
        for (int q = 0; q < 3; q++) { double *t = cur[q]; cur[q] = nxt[q]; nxt[q] = t; }
    }
    OFX_LAUNCH_CHECK(ctx);
    if (launches & 1) {                                   // the result sits in the scratch planes: bring it home (sets that are
        for (int q = 0; q < 3; q++)                       // not iterating were not written there: copy set by set)
            for (int gi = 0; gi < G; gi++)
                if ((grp.mask >> gi) & 1u)
                    OFX_HIP(ctx, hipMemcpyAsync(nxt[q] + (size_t) gi * nx * ny, cur[q] + (size_t) gi * nx * ny,
                                                (size_t) nx * ny * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return OFX_OK;
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
    hipLaunchKernelGGL(k_occ_divu, grid, block, 0, ctx->stream, (const double *) di[0], (const double *) di[1], div_u, nx, ny, OCC_ONE);
    OFX_LAUNCH_CHECK(ctx);
    const OccChi a = {di[0], di[1], di[2], di[3], di[4], di[5], di[6], di[7], di[8], di[9], di[10], di[11], di[12], di[14], di[15],
                      div_u, di[13]};
    double *alt;
    OFX_TRY(ofx_alloc(ctx, 3 * n, &alt));
    OFX_TRY(occ_chi_iterations(ctx, a, di[13], di[14], di[15], alt, nx, ny, n_iter, lambda, theta, alpha, beta, tau_chi, tau_eta, 1, OCC_ONE));
    OFX_TRY(d.out(di[13], chi, n));
    OFX_TRY(d.out(di[14], eta1, n));
    OFX_TRY(d.out(di[15], eta2, n));
    return d.sync();
}

// ---- Scalar_ROF_BoxCellCentered (src/tvl1occflow_tv_rof_box.cpp:22-645) and Solver_wrt_u (tvl1occflow_solvers.cpp:150-216) --------
// nIter times { alfa = |grad u| / (lambda g) per cell;  ONE in-place box-relaxation sweep over the cells in lexicographic
// order;  u = lambda f + lambda div P }.  The sweep is a sequential recurrence: each cell solves the 2x2 / 3x3 / 4x4 system of
// the dual values on its own edges from the edges of its eight neighbours.  Like the SOR sweeps (ofx_sor.hip) it is
// executed on hyperplanes without changing an operand: cell (ci, cj) runs at step q = 2 ci + cj, so that the cells it
// must follow (W, NW, N, NE) are done and the ones it must precede (E, SW, S, SE) are not; the cells of one step touch
// disjoint edges and read nothing another cell of the step writes.
// Cell-centred storage instead of the reference's (2 ny + 1) x (2 nx + 1) staggered grid: Ps / Pe = dual value on the
// south / east edge of a cell (the in/out state initialP1 / initialP2; a cell's north / west edge is the south / east
// edge of its neighbour, 0 on the image border), Fs / Fe = differences of f across those edges (0 on the border), AL =
// alfa of the cell (the reference stores the same value on both edges).
// * Hyperplane-major arrays.  One thread owns one image row, so the 64 lanes of a wave touch 64 different rows at every step;
//   in row-major arrays that is 64 cache lines per load instruction.  All arrays of a sweep are therefore stored with
//   index (2 ci + cj) ny + ci: the cells of one step are contiguous (same layout idea as LaySkew in ofx_sor.hip); Ps / Pe
//   and Fs / Fe are interleaved pairs (one 16-byte access each).
// * A launch works from LDS.  One thread owns one row; a workgroup first brings everything its ROF_K steps will touch into
//   LDS -- the (Ps, Pe) pairs of positions q0 - 4 .. q1 + 2 of the skewed coordinate p = 2 ci + cj and the (Fs, Fe), alfa
//   of positions q0 - 2 .. q1 + 1, per row -- with all loads in flight at once and shared between the three waves so that none
//   exceeds the 63 loads a wave can have outstanding (one memory latency per launch instead of one per step), then runs its
//   steps on LDS only: every neighbour access is an LDS access, one barrier per step, and the thread that makes the LAST
//   update of a value (the north edge from the row below, the west edge from the next cell of the row) also stores it to
//   global memory, without waiting for the store.  Entries whose last update falls into a later launch (2 per row) are
//   written back at the end.  Where a step's time goes: profiles/r02_k_rof_window_breakdown.txt.
// * Rows are cut into blocks of ROF_R = 125, one workgroup each: 128 columns of LDS = 125 own rows + 3 halo rows (two above,
//   one below), walked by 128 threads, plus a third wave for the image's first / last row (see the kernel).  Block b runs ROF_LAG steps behind block b - 1 and a launch executes
//   ROF_K steps of every block; ROF_LAG = ROF_K + 8 guarantees that whatever a workgroup takes from global memory that
//   another one wrote (read up to ROF_K + 2 positions ahead, stored up to 1 step late) was written by an earlier launch,
//   and that the rows below are still untouched by the next block.
#ifndef ROF_VAR
#define ROF_VAR 0                    // 1 .. 5: timing experiments of tools/rof_variants.sh, never shipped
#endif
#define ROF_NT 128
#define ROF_R (ROF_NT - 3)
#define ROF_THREADS (ROF_NT + 64)     // + one wave whose first two threads walk the image's first and last row
#define ROF_K 24
#define ROF_LAG (ROF_K + 8)
// Iterations in flight.  One iteration of the solver is alfa -> sweep -> u, and alfa of the next iteration needs u.  But
// u = lambda f + lambda div P at a cell only needs the dual values around that cell, which are final as soon as the sweep's
// wavefront is four positions past it: the sweep of iteration s therefore follows ROF_LAGI positions behind the sweep of
// iteration s - 1, and between the two wavefronts -- ROF_D positions ahead of sweep s -- an "alfa stage" (workgroups of the
// same launches) computes alfa_s straight from the pairs sweep s - 1 has left behind (u_{s-1} at the cell, its east and
// its south neighbour, never stored).  A call of n iterations is then one chain of qmax + ROF_LAGI (n - 1) steps instead of
// n chains of qmax steps.  The conditions (every value an alfa stage or a sweep takes from another workgroup was stored by an
// earlier launch and is overwritten by a later one) are enumerated by tools/check_rof_pipeline.py: D in [58, 60], LAGI >= 118;
// an image of one row block (ny <= ROF_R: the coarse pyramid levels, where the pipeline's depth is most of the chain) has no
// block lag to respect: D in [26, 28], LAGI >= 54.
#define ROF_D (ROF_K + ROF_LAG + 2)
#define ROF_LAGI 120
#define ROF_D_1 (ROF_K + 2)
#define ROF_LAGI_1 56
// The window length K is a template parameter of the kernel: ROF_KG = 10 steps (75 KB of LDS, two workgroups per CU) is what
// runs; 24 steps (143 KB, one workgroup per CU; the round-2 geometry that the text above describes) remains as option
// "rof_window" = 24.  Everything derived from K follows the same formulas (tools/check_rof_pipeline.py
// checks both parameter sets): blocks K + 8 apart, alfa stage K + LAG + 2 (K + 2 for one row block) ahead of its sweep, sweeps
// D + K + LAG + 6 (D + K + 6) apart.
#define ROF_KG 10
template <int K> struct RofGeo {
    static constexpr int LAG = K + 8;
    static constexpr int RING = K + 7;       // positions q0 - 4 .. q1 + 2
    static constexpr int COEF = K + 3;       // positions q0 - 2 .. q1 + 1
    static constexpr size_t LDS = (size_t) ROF_NT * (RING * sizeof(double2) + COEF * (sizeof(double2) + sizeof(double)));
    static constexpr int D = K + LAG + 2, LAGI = D + K + LAG + 6;      // several row blocks
    static constexpr int D1 = K + 2, LAGI1 = D1 + K + 6;               // one row block
};
static_assert(RofGeo<ROF_K>::LAG == ROF_LAG && RofGeo<ROF_K>::D == ROF_D && RofGeo<ROF_K>::LAGI == ROF_LAGI &&
              RofGeo<ROF_K>::D1 == ROF_D_1 && RofGeo<ROF_K>::LAGI1 == ROF_LAGI_1, "the documented lags of the 24-step window");
struct RofArr {                      // all arrays hyperplane-major; PP = (Ps, Pe) and FF = (Fs, Fe) per cell: one 16-byte access each
    double2 *PP;
    const double2 *FF;
    const double *AL;
    int nx, ny;
};
static inline __host__ __device__ size_t rof_skew_elems(int nx, int ny) { return (size_t) (2 * (ny - 1) + nx) * ny; }
OFX_DEV size_t rof_sk(int ci, int cj, int ny) { return (size_t) (2 * ci + cj) * ny + ci; }

// the LDS copy of one workgroup's launch window and what a cell may do with it
struct RofRing {
    const RofArr &a;
    double2 (*win)[ROF_NT];          // [position - (q0 - 4)][column]: .x = Ps, .y = Pe
    double2 (*cff)[ROF_NT];          // [position - (q0 - 2)][column]: .x = Fs, .y = Fe
    double (*cal)[ROF_NT];           // [position - (q0 - 2)][column]: alfa
    int row0;                        // image row of column 0 (= first own row - 2)
    int pw0;                         // q0 - 4
    bool keep_s;                     // this row's south edges are final for this workgroup (last own row): store them
    OFX_DEV double2 &at(int ci, int cj) const { return win[2 * ci + cj - pw0][ci - row0]; }
    OFX_DEV double ps(int ci, int cj) const { return (ci >= 0 && cj >= 0) ? at(ci, cj).x : 0.0; }
    OFX_DEV double pe(int ci, int cj) const { return (ci >= 0 && cj >= 0) ? at(ci, cj).y : 0.0; }
    OFX_DEV double Fs(int ci, int cj) const { return cff[2 * ci + cj - pw0 - 2][ci - row0].x; }
    OFX_DEV double Fe(int ci, int cj) const { return cff[2 * ci + cj - pw0 - 2][ci - row0].y; }
    OFX_DEV double AL(int ci, int cj) const { return cal[2 * ci + cj - pw0 - 2][ci - row0]; }
#if ROF_VAR == 2                     // timing experiment: no global stores from the steps
    OFX_DEV void put_w(int ci, int cj, double v) const { at(ci, cj - 1).y = v; }
    OFX_DEV void put_n(int ci, int cj, double v) const { at(ci - 1, cj).x = v; }
#else
    OFX_DEV void put_w(int ci, int cj, double v) const { at(ci, cj - 1).y = v; a.PP[rof_sk(ci, cj - 1, a.ny)].y = v; }   // final
    OFX_DEV void put_n(int ci, int cj, double v) const { at(ci - 1, cj).x = v; a.PP[rof_sk(ci - 1, cj, a.ny)].x = v; }   // final
#endif
    OFX_DEV void put_s(int ci, int cj, double v) const
    {
        at(ci, cj).x = v;
        if (keep_s) a.PP[rof_sk(ci, cj, a.ny)].x = v;
    }
    OFX_DEV void put_e(int ci, int cj, double v) const { at(ci, cj).y = v; }
};

// What an inner cell computes from the alfa values alone (its Gauss elimination factors and the four denominators).  It is
// taken off the sequential chain: while a thread walks the chain of cell (ci, cj) -- four dependent divisions once these
// are known -- the same basic block computes the factors of cell (ci, cj + 1), which fills the chain's latency bubbles.
struct RofPre {
    double aa, bb, alf, gam, cc, d1, d2, d3, b0;
};
OFX_DEV RofPre rof_pre(double b0, double b1, double b2, double b3)
{
    RofPre p;
    p.aa = 1 / b0;
    p.bb = -(b0 + 1) / (b0 * b1 - 1);
    p.alf = 1 + p.aa;
    p.gam = -p.aa + p.bb * p.alf;
    p.cc = (1 - p.gam) / (b2 + p.gam);
    p.d1 = (b3 + p.gam + p.cc * (p.gam - 1));
    p.d2 = (b2 + p.gam);
    p.d3 = (b1 - p.aa);
    p.b0 = b0;
    return p;
}

// one cell; the nine kinds keep the reference's own closed forms and association order (only the north side writes its
// free terms with the F term first).  Every LDS operand of every kind is read up front, unconditionally -- all slots exist,
// the ones of cells outside the image hold don't-care values that the selects below replace by 0 -- so that a step pays one LDS
// round trip, not one per neighbour.
OFX_DEV void rof_cell(const RofRing &r, int ci, int cj, double w, RofPre &pre, bool &pre_ok)
{
    const RofArr &a = r.a;
    const int nx = a.nx, ny = a.ny;
    const bool top = ci == 0, bot = ci == ny - 1, lef = cj == 0, rig = cj == nx - 1;
    const double2 c_m2 = r.at(ci, cj - 2), c_m1 = r.at(ci, cj - 1), c_0 = r.at(ci, cj), c_p1 = r.at(ci, cj + 1);
    const double2 n_m1 = r.at(ci - 1, cj - 1), n_0 = r.at(ci - 1, cj), n_p1 = r.at(ci - 1, cj + 1), nn_0 = r.at(ci - 2, cj);
    const double2 s_0 = r.at(ci + 1, cj), s_m1 = r.at(ci + 1, cj - 1);
    const double fe_w = r.Fe(ci, cj - 1), fe_c = r.Fe(ci, cj), fs_n = r.Fs(ci - 1, cj), fs_c = r.Fs(ci, cj);
    const double al_w = r.AL(ci, cj - 1), al_n = r.AL(ci - 1, cj), al = r.AL(ci, cj);
    const double al_e = r.AL(ci, cj + 1), al_ne = r.AL(ci - 1, cj + 1);      // for the factors of the next cell of the row
    // keep the compiler from sinking some of these reads into the conditional blocks below (one more round trip each)
    asm volatile("" ::"v"(c_m2.y), "v"(c_m1.x), "v"(c_m1.y), "v"(c_0.x), "v"(c_0.y), "v"(c_p1.x), "v"(c_p1.y), "v"(n_m1.x), "v"(n_m1.y),
                 "v"(n_0.x), "v"(n_0.y), "v"(n_p1.x), "v"(nn_0.x), "v"(s_0.x), "v"(s_0.y), "v"(s_m1.y));
    asm volatile("" ::"v"(fe_w), "v"(fe_c), "v"(fs_n), "v"(fs_c), "v"(al_w), "v"(al_n), "v"(al), "v"(al_e), "v"(al_ne));
#if ROF_VAR == 5                     // timing experiment: the LDS reads of a step and nothing else
    return;
#endif
    const bool c1 = cj >= 1, c2 = cj >= 2, r1 = ci >= 1, r2 = ci >= 2;         // does the neighbour exist (else the value is 0)
    const bool nside = top && !lef && !rig;
    double W = 0, N = 0, S = 0, E = 0;
    if (!lef) {
        const double pe_m2 = c2 ? c_m2.y : 0.0, ps_m1 = c_m1.x, ps_nm1 = r1 ? n_m1.x : 0.0;
        W = nside ? -fe_w - pe_m2 + ps_m1 - ps_nm1 : -pe_m2 + ps_m1 - ps_nm1 - fe_w;
    }
    if (!top) N = -(r2 ? nn_0.x : 0.0) + n_0.y - (c1 ? n_m1.y : 0.0) - fs_n;
    if (!bot) {
        const double pe_sm1 = c1 ? s_m1.y : 0.0;
        S = nside ? -fs_c - s_0.x - s_0.y + pe_sm1 : -s_0.x - s_0.y + pe_sm1 - fs_c;
    }
    if (!rig) {
        const double ps_np1 = r1 ? n_p1.x : 0.0;
        E = nside ? -fe_c - c_p1.y - c_p1.x + ps_np1 : -c_p1.y - c_p1.x + ps_np1 - fe_c;
    }
    const double b0 = lef ? 0.0 : -2 - al_w, b1 = top ? 0.0 : -2 - al_n;
    const double b2 = bot ? 0.0 : -2 - al, b3 = rig ? 0.0 : -2 - al;
    // own edges: west = east edge of the left cell, north = south edge of the cell above
    const double ow = lef ? 0.0 : c_m1.y, on = top ? 0.0 : n_0.x, os = c_0.x, oe = c_0.y;
    double den;
    if (top && lef) {
        den = b2 * b3 - 1;
        const double s_ = (1 - w) * os + w * (S * b3 + E) / den, e_ = (1 - w) * oe + w * (E * b2 + S) / den;
        r.put_s(ci, cj, s_); r.put_e(ci, cj, e_);
    } else if (top && rig) {
        den = b0 * b2 - 1;
        const double w_ = (1 - w) * ow + w * (W * b2 - S) / den, s_ = (1 - w) * os + w * (S * b0 - W) / den;
        r.put_w(ci, cj, w_); r.put_s(ci, cj, s_);
    } else if (top) {
        den = b0 * b2 * b3 - b0 - b2 - b3 - 2;
        const double w_ = (1 - w) * ow + w * (W * b2 * b3 - E * b2 - S * b3 - W - E - S) / den;
        const double s_ = (1 - w) * os + w * (S * b0 * b3 - W * b3 + E * b0 - W + E - S) / den;
        const double e_ = (1 - w) * oe + w * (E * b0 * b2 - W * b2 + S * b0 - W - E + S) / den;
        r.put_w(ci, cj, w_); r.put_s(ci, cj, s_); r.put_e(ci, cj, e_);
    } else if (bot && lef) {
        den = b3 * b1 - 1;
        const double n_ = (1 - w) * on + w * (b3 * N - E) / den, e_ = (1 - w) * oe + w * (b1 * E - N) / den;
        r.put_n(ci, cj, n_); r.put_e(ci, cj, e_);
    } else if (bot && rig) {
        den = b0 * b1 - 1;
        const double w_ = (1 - w) * ow + w * (W * b1 + N) / den, n_ = (1 - w) * on + w * (N * b0 + W) / den;
        r.put_w(ci, cj, w_); r.put_n(ci, cj, n_);
    } else if (bot) {
        den = b0 * b1 * b3 - b0 - b1 - b3 - 2;
        const double w_ = (1 - w) * ow + w * (W * b1 * b3 - E + N - E * b1 - W + N * b3) / den;
        const double n_ = (1 - w) * on + w * (N * b0 * b3 + W - E - N - E * b0 + W * b3) / den;
        const double e_ = (1 - w) * oe + w * (E * b0 * b1 - N - W - W * b1 - N * b0 - E) / den;
        r.put_w(ci, cj, w_); r.put_n(ci, cj, n_); r.put_e(ci, cj, e_);
    } else if (lef) {
        den = b1 * b2 * b3 - (b1 + b2 + b3) - 2;
        const double n_ = (1 - w) * on + w * (b2 * b3 * N - E * b2 - S * b3 - N - S - E) / den;
        const double s_ = (1 - w) * os + w * (b1 * b3 * S + E * b1 - N * b3 - N - S + E) / den;
        const double e_ = (1 - w) * oe + w * (b1 * b2 * E - N * b2 + S * b1 - N + S - E) / den;
        r.put_n(ci, cj, n_); r.put_s(ci, cj, s_); r.put_e(ci, cj, e_);
    } else if (rig) {
        den = (b0 * b1 * b2) + (-b0 - b1 - b2 - 2);
        const double w_ = (1 - w) * ow + w * (W * b1 * b2 - S + N - S * b1 - W + N * b2) / den;
        const double n_ = (1 - w) * on + w * (N * b0 * b2 + W - S - N - S * b0 + W * b2) / den;
        const double s_ = (1 - w) * os + w * (S * b0 * b1 - N - W - W * b1 - N * b0 - S) / den;
        r.put_w(ci, cj, w_); r.put_n(ci, cj, n_); r.put_s(ci, cj, s_);
    } else {                                             // inner cell: Gauss elimination, each value from the new ones before it
#if ROF_VAR == 1                     // timing experiment: no arithmetic to speak of
        r.put_e(ci, cj, oe + E); r.put_s(ci, cj, os + S); r.put_n(ci, cj, on + N); r.put_w(ci, cj, ow + W + b0 + b1 + b2 + b3);
        return;
#endif
        RofPre P = pre;
        if (!pre_ok) P = rof_pre(b0, b1, b2, b3);        // first inner cell of the row / of the launch
        const double x = N + P.aa * W;
        const double y = -P.aa * W + P.bb * x;
        const double e_ = (1 - w) * oe + w * (E + y + P.cc * (S + y)) / P.d1;
        const double s_ = (1 - w) * os + w * (S + y + e_ * (1 - P.gam)) / P.d2;
        const double n_ = (1 - w) * on + w * (x - P.alf * (e_ + s_)) / P.d3;
        const double w_ = (1 - w) * ow + w * (W + n_ - s_ - e_) / (P.b0);
        r.put_e(ci, cj, e_); r.put_s(ci, cj, s_); r.put_n(ci, cj, n_); r.put_w(ci, cj, w_);
        pre = rof_pre(-2 - al, -2 - al_ne, -2 - al_e, -2 - al_e);      // cell (ci, cj + 1): b0 from this cell's alfa
        pre_ok = cj + 1 < nx - 1;
        return;
    }
    pre_ok = false;
}

// ROF_K steps of every row block of one sweep, launch window starting at global step T0; blockIdx.y = which of the
// (independent) problems
struct RofSet {
    RofArr a[2];
    const double *LF[2];             // lambda f per cell and problem, hyperplane-major (alfa stage)
    const double *LG;                // lambda g per cell, hyperplane-major (shared by the problems of a set)
    double *ALw[2];                  // a[k].AL, writable (alfa stage)
    size_t stride;                   // lockstep groups: problem set g = blockIdx.z works on arrays offset by g * stride
    unsigned mask;                   // bit g: still iterating
    int nc, B;                       // problems per set, row blocks: blockIdx.y = problem + nc * iteration, blockIdx.x = block (+ B: alfa stage)
    int lagi, d;                     // positions between the sweeps of consecutive iterations / alfa stage ahead of its sweep
};
// u = lambda f + lambda (P_south - P_north + P_east - P_west) of one cell from the hyperplane-major pairs (k_rof_u's expression)
OFX_DEV double rof_u_cell(const double2 *__restrict__ PP, const double *__restrict__ LF, int ci, int cj, int ny, double lambda)
{
    const size_t k = rof_sk(ci, cj, ny);
    const double2 own = PP[k];
    const double pn = ci > 0 ? PP[rof_sk(ci - 1, cj, ny)].x : 0.0, pw = cj > 0 ? PP[rof_sk(ci, cj - 1, ny)].y : 0.0;
    return LF[k] + lambda * (own.x - pn + own.y - pw);
}
// alfa stage of one row block: alfa of the cells of its rows at positions [p0, p0 + ROF_K) (k_rof_alfa's expression on
// u recomputed from the pairs).  No dependence between cells: consecutive threads take consecutive rows of one position.
OFX_DEV void rof_alfa_band(const RofArr &a, const double *__restrict__ LF, const double *__restrict__ LG, double *__restrict__ AL,
                           int b, int p0, double lambda, int K)
{
    const int nx = a.nx, ny = a.ny;
    for (int idx = (int) threadIdx.x; idx < ROF_R * K; idx += ROF_THREADS) {
        const int ci = b * ROF_R + idx % ROF_R, p = p0 + idx / ROF_R;
        const int cj = p - 2 * ci;
        if (ci >= ny || cj < 0 || cj >= nx) continue;
        const double u0 = rof_u_cell(a.PP, LF, ci, cj, ny, lambda);
        const double ux = (cj < nx - 1) ? rof_u_cell(a.PP, LF, ci, cj + 1, ny, lambda) - u0 : 0.0;
        const double uy = (ci < ny - 1) ? rof_u_cell(a.PP, LF, ci + 1, cj, ny, lambda) - u0 : 0.0;
        const size_t k = rof_sk(ci, cj, ny);
        AL[k] = sqrt(ux * ux + uy * uy) / LG[k];
    }
}
template <int K>
__global__ __launch_bounds__(ROF_THREADS) void k_rof_window(RofSet s, int T0, double w, double lambda)
{
    constexpr int ROF_RING = RofGeo<K>::RING, ROF_COEF = RofGeo<K>::COEF, LAG = RofGeo<K>::LAG;
    extern __shared__ double2 rof_lds[];
    double2 (*win)[ROF_NT] = reinterpret_cast<double2 (*)[ROF_NT]>(rof_lds);
    double2 (*cff)[ROF_NT] = reinterpret_cast<double2 (*)[ROF_NT]>(rof_lds + ROF_RING * ROF_NT);
    double (*cal)[ROF_NT] = reinterpret_cast<double (*)[ROF_NT]>(rof_lds + (ROF_RING + ROF_COEF) * ROF_NT);
    if (!((s.mask >> blockIdx.z) & 1u)) return;
    const int it = (int) blockIdx.y / s.nc, prob = (int) blockIdx.y - it * s.nc;    // iteration in flight, problem of the set
    const bool alfa_stage = (int) blockIdx.x >= s.B;
    const int b = (int) blockIdx.x - (alfa_stage ? s.B : 0);
    RofArr a = prob ? s.a[1] : s.a[0];                                      // by value: a dynamic index would be re-read from the kernel arguments at every use
    const size_t off = (size_t) blockIdx.z * s.stride;
    a.PP += off; a.FF += off; a.AL += off;
    const int nx = a.nx, ny = a.ny;
    const int t = (int) threadIdx.x, row0 = b * ROF_R - 2;
    const int q0 = T0 - LAG * b - s.lagi * it, q1 = q0 + K - 1;
    // positions at which the block's own rows have cells: outside them the unit has nothing to do (uniform over the workgroup)
    const int p_lo = 2 * b * ROF_R, p_hi = 2 * (min(b * ROF_R + ROF_R, ny) - 1) + nx - 1;
    // The alfa stage of (iteration, block) has workgroups of its own.  (Run as the head of the sweep's workgroup instead -- half
    // the workgroups per launch, each of which fills a CU's LDS -- it changes nothing for lockstep groups, 5.4 ms per 640x480
    // triple either way, and costs a lone solve 7 %: measured, dropped.)
    if (alfa_stage) {
        if (it == 0 || q0 + s.d > p_hi || q1 + s.d < p_lo) return;          // alfa of iteration 0 comes from the seed (k_rof_alfa)
        rof_alfa_band(a, (prob ? s.LF[1] : s.LF[0]) + off, s.LG + off, (prob ? s.ALw[1] : s.ALw[0]) + off, b, q0 + s.d, lambda, K);
        return;
    }
    if (q0 > p_hi || q1 < p_lo) return;
    // Who walks which row.  Threads 0 .. 127 = LDS columns = rows row0 .. row0 + 127 (125 own rows between two halo rows above
    // and one below).  The image's first and last row consist of cells of another kind (3 x 3 systems) than the inner rows:
    // left in their column's wave they would make that wave -- and with it the whole launch, which waits for its slowest
    // workgroup -- execute two kinds per step.  They are walked by two threads of a third wave instead.
    int ci = row0 + t;
    const bool column = t < ROF_NT;
    const bool row_own = column && ci >= 0 && ci < ny && t >= 2 && t < 2 + ROF_R;        // this column's row belongs to the block
    bool own = row_own && ci != 0 && ci != ny - 1;
    if (!column) {
        ci = (t == ROF_NT) ? 0 : ny - 1;
        own = t < ROF_NT + 2 && ci - row0 >= 2 && ci - row0 < 2 + ROF_R;
    }
    const RofRing ring = {a, win, cff, cal, row0, q0 - 4, ci - row0 == 1 + ROF_R};
    // Everything the launch reads.  Unconditional loads (out-of-range entries read a harmless address and are never used)
    // into registers first, so that all of them are in flight together -- one memory latency, not one per entry -- and split so
    // that no wave has more loads than fit in flight at once (63): the column threads bring their (Ps, Pe) and (Fs, Fe)
    // pairs, 31 + 27 16-byte loads; the third wave brings alfa for two columns per lane, 2 x 27 8-byte loads.
    if (column) {
        const int cic = min(max(ci, 0), ny - 1);
        double2 ring_in[ROF_RING], ff_in[ROF_COEF];
#pragma unroll
        for (int k = 0; k < ROF_RING; k++) {
            const int p = q0 - 4 + k, x = p - 2 * cic;
            ring_in[k] = a.PP[(x >= 0 && x < nx) ? (size_t) p * ny + cic : (size_t) cic];
        }
#pragma unroll
        for (int k = 0; k < ROF_COEF; k++) {
            const int p = q0 - 2 + k, x = p - 2 * cic;
            ff_in[k] = a.FF[(x >= 0 && x < nx) ? (size_t) p * ny + cic : (size_t) cic];
        }
#pragma unroll
        for (int k = 0; k < ROF_RING; k++) win[k][t] = ring_in[k];
#pragma unroll
        for (int k = 0; k < ROF_COEF; k++) cff[k][t] = ff_in[k];
    } else {
        const int lane = t - ROF_NT;                       // 0 .. 63: columns lane and lane + 64
        double al_in[2][ROF_COEF];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int cc = min(max(row0 + lane + 64 * h, 0), ny - 1);
#pragma unroll
            for (int k = 0; k < ROF_COEF; k++) {
                const int p = q0 - 2 + k, x = p - 2 * cc;
                al_in[h][k] = a.AL[(x >= 0 && x < nx) ? (size_t) p * ny + cc : (size_t) cc];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k = 0; k < ROF_COEF; k++) cal[k][lane + 64 * h] = al_in[h][k];
    }
    __syncthreads();
    RofPre pre = {};
    bool pre_ok = false;
    for (int q = q0; q <= q1; q++) {
        const int cj = q - 2 * ci;
#if ROF_VAR != 4                     // 4: timing experiment, the step loop without its cells
        if (own && cj >= 0 && cj < nx) rof_cell(ring, ci, cj, w, pre, pre_ok);
#endif
#if ROF_VAR == 3                     // timing experiment: no barrier between the steps (results are wrong)
        __builtin_amdgcn_s_waitcnt(0xc07f);
#else
        __syncthreads();
#endif
    }
    if (row_own) {                                                           // entries a later launch still has to update
        for (int p = q1 - 1; p <= q1; p++) {
            const int x = p - 2 * ci;
            if (x >= 0 && x < nx) {
                const size_t e = (size_t) p * ny + ci;
                a.PP[e].x = win[p - q0 + 4][t].x;
                if (p == q1) a.PP[e].y = win[p - q0 + 4][t].y;
            }
        }
    }
}

// row-major planes (Ps, Pe) <-> hyperplane-major pairs (state of the host-facing entry points)
__global__ void k_rof_skew(double *__restrict__ ps, double *__restrict__ pe, double2 *__restrict__ pp, int nx, int ny, int to_skew)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t c = (size_t) i * nx + j, k = rof_sk(i, j, ny);
    if (to_skew) pp[k] = make_double2(ps[c], pe[c]);
    else {
        const double2 v = pp[k];
        ps[c] = v.x;
        pe[c] = v.y;
    }
}
// edge differences of f (once per call), :137-164
// ... and the two products the alfa stage needs per cell, in the sweep's layout: lambda f (the first term of u) and lambda g
// (the divisor of alfa); LG may be null (second problem of a set: same g)
__global__ void k_rof_fdiff(const double *__restrict__ f, double2 *__restrict__ FF, int nx, int ny, OccGrp Grm, size_t sk_stride,
                            const double *__restrict__ g, double *__restrict__ LF, double *__restrict__ LG, double lambda)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(Grm, o)) return;
    const size_t c = o + (size_t) i * nx + j, k = blockIdx.z * sk_stride + rof_sk(i, j, ny);
    FF[k] = make_double2((i < ny - 1) ? f[c + nx] - f[c] : 0.0, (j < nx - 1) ? f[c + 1] - f[c] : 0.0);
    LF[k] = lambda * f[c];
    if (LG) LG[k] = lambda * g[c];
}
// alfa = hypot(forward gradient of u) / (lambda g) with the file-local hypot = sqrt(x x + y y), :15-20,173-187
struct RofPt {
    const double *u[2], *f[2];
    const double2 *PP[2];
    double *AL[2], *uo[2];
    int nc;                          // problems per set; blockIdx.z = set * nc + problem
    size_t rm_stride, sk_stride;     // per set: row-major planes (u, f, g), hyperplane-major planes (Ps, Pe, AL)
    unsigned mask;
};
__global__ void k_rof_alfa(RofPt a, const double *__restrict__ g, int nx, int ny, double lambda)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z % a.nc, set = blockIdx.z / a.nc;
    if (j >= nx || i >= ny || !((a.mask >> set) & 1u)) return;
    const size_t c = set * a.rm_stride + (size_t) i * nx + j;
    const double *u = a.u[k];
    const double ux = (j < nx - 1) ? u[c + 1] - u[c] : 0.0, uy = (i < ny - 1) ? u[c + nx] - u[c] : 0.0;
    a.AL[k][set * a.sk_stride + rof_sk(i, j, ny)] = sqrt(ux * ux + uy * uy) / (lambda * g[c]);
}
// u = lambda f + lambda (P_south - P_north + P_east - P_west), :616-640
__global__ void k_rof_u(RofPt a, int nx, int ny, double lambda)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z % a.nc, set = blockIdx.z / a.nc;
    if (j >= nx || i >= ny || !((a.mask >> set) & 1u)) return;
    const size_t c = set * a.rm_stride + (size_t) i * nx + j;
    const double2 *PP = a.PP[k] + set * a.sk_stride;
    const double2 own = PP[rof_sk(i, j, ny)];
    const double pn = i > 0 ? PP[rof_sk(i - 1, j, ny)].x : 0.0, pw = j > 0 ? PP[rof_sk(i, j - 1, ny)].y : 0.0;
    a.uo[k][c] = lambda * a.f[k][c] + lambda * (own.x - pn + own.y - pw);
}

// the launches of one Scalar_ROF_BoxCellCentered call with windows of K steps (see rof_box_dev)
template <int K>
static int rof_box_windows(ofx_ctx *ctx, RofSet set, const RofPt &pt, const double *g, dim3 grid, dim3 block, int nc, int G, int B,
                           int qmax, int n_iter, double omega, double lambda)
{
    using Geo = RofGeo<K>;
    const int nx = set.a[0].nx, ny = set.a[0].ny;
    set.lagi = B > 1 ? Geo::LAGI : Geo::LAGI1;
    set.d = B > 1 ? Geo::D : Geo::D1;
    static std::atomic<unsigned> lds_set(0);           // bit d: the attribute has been set on device d (per device, any thread; per K)
    if (!(lds_set.load() & (1u << (ctx->device & 31)))) {
        OFX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_rof_window<K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int) Geo::LDS));
        lds_set.fetch_or(1u << (ctx->device & 31));
    }
    if (ctx->rof_pipe) {
        const long total = (long) qmax + 1 + (long) Geo::LAG * (B - 1) + (long) set.lagi * (n_iter - 1);
        hipLaunchKernelGGL(k_rof_alfa, grid, block, 0, ctx->stream, pt, g, nx, ny, lambda);
        for (long T0 = 0; T0 < total; T0 += K) {
            // the grid covers iterations 0 .. it_hi (the last one whose alfa stage has reached position 0); units outside
            // their range of positions leave at once
            long it_hi = (T0 + K + set.d) / set.lagi;
            if (it_hi > n_iter - 1) it_hi = n_iter - 1;
            hipLaunchKernelGGL(k_rof_window<K>, dim3(2 * B, nc * (int) (it_hi + 1), G), dim3(ROF_THREADS), Geo::LDS, ctx->stream, set,
                               (int) T0, omega, lambda);
        }
        hipLaunchKernelGGL(k_rof_u, grid, block, 0, ctx->stream, pt, nx, ny, lambda);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    }
    const long total = (long) qmax + 1 + (long) Geo::LAG * (B - 1);
    for (int it = 0; it < n_iter; it++) {
        hipLaunchKernelGGL(k_rof_alfa, grid, block, 0, ctx->stream, pt, g, nx, ny, lambda);
        for (long T0 = 0; T0 < total; T0 += K)
            hipLaunchKernelGGL(k_rof_window<K>, dim3(B, nc, G), dim3(ROF_THREADS), Geo::LDS, ctx->stream, set, (int) T0, omega, lambda);
        hipLaunchKernelGGL(k_rof_u, grid, block, 0, ctx->stream, pt, nx, ny, lambda);
        OFX_LAUNCH_CHECK(ctx);
    }
    return OFX_OK;
}

// nc = 1 | 2 independent problems sharing g, lambda and the size (the two flow components of Solver_wrt_u), every launch
// serving both -- and all G sets of them (lockstep groups: set s on planes offset by s * nx * ny / s * rof_skew_elems(),
// sets whose bit of `mask` is clear are left alone).  Device arrays in place: u[k] (in: seed, out: result; row-major),
// PP[k] = (Ps, Pe) pairs (in/out state, HYPERPLANE-MAJOR, rof_skew_elems() pairs per set); scratch = ROF_SCRATCH_PLANES(nc) G
// hyperplane-major planes of doubles (the (Fs, Fe) pairs of every problem first, then alfa, lambda f, and one plane lambda g).
// All n_iter iterations are in flight together (see ROF_LAGI): the seed's alfa, one chain of windows, the final u.
// Option "rof_pipe" = 0: one iteration at a time (alfa pass, windows, u pass per iteration), the round-2 schedule.
#define ROF_SCRATCH_PLANES(nc) (4 * (nc) + 1)
static int rof_box_dev(ofx_ctx *ctx, int nc, double *const *u, const double *const *f, double2 *const *PP, const double *g,
                       double lambda, double omega, int nx, int ny, int n_iter, double *scratch, int G = 1, unsigned mask = 1u)
{
    const size_t n = rof_skew_elems(nx, ny), nrm = (size_t) nx * ny;
    const dim3 grid(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), nc * G), block(64, 4);
    RofSet set;
    RofPt pt;
    double *LG = scratch + (size_t) (4 * nc) * n * G;
    for (int k = 0; k < 2; k++) {
        const int c = k < nc ? k : 0;
        double2 *FF = reinterpret_cast<double2 *>(scratch + (2 * c) * n * G);        // pairs first: 16-byte aligned whatever n
        double *AL = scratch + (2 * nc + c) * n * G, *LF = scratch + (3 * nc + c) * n * G;
        if (k < nc) {
            hipLaunchKernelGGL(k_rof_fdiff, dim3(grid.x, grid.y, G), block, 0, ctx->stream, f[c], FF, nx, ny, OccGrp{nrm, mask}, n, g, LF,
                               k == 0 ? LG : (double *) nullptr, lambda);
            OFX_LAUNCH_CHECK(ctx);
        }
        set.a[k] = RofArr{PP[c], FF, AL, nx, ny};
        set.LF[k] = LF; set.ALw[k] = AL;
        pt.u[k] = u[c]; pt.f[k] = f[c]; pt.PP[k] = PP[c]; pt.AL[k] = AL; pt.uo[k] = u[c];
    }
    const int B = ofx_cdiv(ny, ROF_R), qmax = 2 * (ny - 1) + nx - 1;
    set.LG = LG; set.stride = n; set.mask = mask; set.nc = nc; set.B = B;
    pt.nc = nc; pt.rm_stride = nrm; pt.sk_stride = n; pt.mask = mask;
    if (n_iter < 1) return OFX_OK;
    // window length: 10 steps per launch unless option "rof_window" = 24.  Measured (profiles/r03_x_rof_window_length.txt; 6 / 8 / 10 /
    // 12 / 16 / 24 steps): one 640x480 triple 21.7 / 21.4 / 21.4 / 22.3 / 23.2 / 26.0 ms, 320x240 12.9 / 12.7 / 12.9 / 13.4 / 14.2 /
    // 16.2 ms -- with all iterations in flight the pipeline's depth (9 LAGI, LAGI ~ 4 K) outweighs the per-launch fill a long
    // window amortises -- and batches of 32 5.1 ms per triple against 5.7 (two 75 KB windows per CU instead of one of 143 KB).
    if (ctx->rof_window == ROF_K) return rof_box_windows<ROF_K>(ctx, set, pt, g, grid, block, nc, G, B, qmax, n_iter, omega, lambda);
    return rof_box_windows<ROF_KG>(ctx, set, pt, g, grid, block, nc, G, B, qmax, n_iter, omega, lambda);
}
// host-facing state planes: upload both row-major planes, interleave; split, download.  tmp = 2 nx ny doubles
static int rof_state_in(ofx_ctx *ctx, const double *hps, const double *hpe, double2 **dev, double *tmp, int nx, int ny)
{
    const size_t n = (size_t) nx * ny;
    OFX_TRY(ofx_alloc(ctx, rof_skew_elems(nx, ny), dev));
    OFX_HIP(ctx, hipMemcpyAsync(tmp, hps, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(tmp + n, hpe, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rof_skew, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)), dim3(64, 4), 0, ctx->stream, tmp, tmp + n, *dev, nx, ny, 1);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
static int rof_state_out(ofx_ctx *ctx, double2 *dev, double *hps, double *hpe, double *tmp, int nx, int ny)
{
    const size_t n = (size_t) nx * ny;
    hipLaunchKernelGGL(k_rof_skew, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)), dim3(64, 4), 0, ctx->stream, tmp, tmp + n, dev, nx, ny, 0);
    OFX_LAUNCH_CHECK(ctx);
    OFX_HIP(ctx, hipMemcpyAsync(hps, tmp, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(hpe, tmp + n, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));      // tmp may be reused by the caller
    return OFX_OK;
}

extern "C" int ofx_scalar_rof_box_cell_centered(ofx_ctx *ctx, double *u, const double *f, double *initialP1, double *initialP2,
                                                const double *g_function, double lambda, double omega, int nx, int ny,
                                                int nIter)
{
    OFX_ENTER(ctx);
    if (!u || !f || !initialP1 || !initialP2 || !g_function) return ofx_fail(ctx, OFX_ERR_ARG, "rof_box: NULL pointer");
    if (nx < 2 || ny < 2 || (long long) nx * ny > 0x3fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "rof_box: bad size %dx%d", nx, ny);
    if (nIter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "rof_box: nIter=%d", nIter);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *du, *df, *dg, *tmp, *scratch;
    double2 *dpp;
    OFX_TRY(d.in(u, &du, n));
    OFX_TRY(d.in(f, &df, n));
    OFX_TRY(d.in(g_function, &dg, n));
    OFX_TRY(ofx_alloc(ctx, 2 * n, &tmp));
    OFX_TRY(rof_state_in(ctx, initialP1, initialP2, &dpp, tmp, nx, ny));
    OFX_TRY(ofx_alloc(ctx, ROF_SCRATCH_PLANES(1) * rof_skew_elems(nx, ny), &scratch));
    const double *fs[1] = {df};
    OFX_TRY(rof_box_dev(ctx, 1, &du, fs, &dpp, dg, lambda, omega, nx, ny, nIter, scratch));
    OFX_TRY(d.out(du, u, n));
    OFX_TRY(rof_state_out(ctx, dpp, initialP1, initialP2, tmp, nx, ny));
    return d.sync();
}

#define OCC_OMEGA 1.25             // src/tvl1occflow_constants.h:28
// f = v / theta + beta grad(chi), u = v + theta beta grad(chi) (tvl1occflow_solvers.cpp:192-203)
__global__ void k_occ_u_init(const double *__restrict__ v1, const double *__restrict__ v2, const double *__restrict__ chi,
                             double *__restrict__ f1, double *__restrict__ f2, double *__restrict__ u1, double *__restrict__ u2,
                             int nx, int ny, double theta, double beta, OccGrp G)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(G, o)) return;
    const size_t c = o + (size_t) i * nx + j;
    const double chix = (j < nx - 1) ? chi[c + 1] - chi[c] : 0.0, chiy = (i < ny - 1) ? chi[c + nx] - chi[c] : 0.0;
    f1[c] = v1[c] / theta + beta * chix;
    f2[c] = v2[c] / theta + beta * chiy;
    u1[c] = v1[c] + theta * beta * chix;
    u2[c] = v2[c] + theta * beta * chiy;
}

extern "C" int ofx_solver_wrt_u(ofx_ctx *ctx, double *u1, double *u2, const double *v1, const double *v2, const double *chi,
                                const double *g, double theta, double beta, int nx, int ny, double *p11, double *p12,
                                double *p21, double *p22, int n_iter)
{
    OFX_ENTER(ctx);
    if (!u1 || !u2 || !v1 || !v2 || !chi || !g || !p11 || !p12 || !p21 || !p22) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_u: NULL pointer");
    if (nx < 2 || ny < 2 || (long long) nx * ny > 0x3fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_u: bad size %dx%d", nx, ny);
    if (n_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "solver_wrt_u: n_iter=%d", n_iter);
    Dev d{ctx};
    const size_t n = (size_t) nx * ny;
    double *dv1, *dv2, *dchi, *dg, *f1, *f2, *du1, *du2, *tmp, *scratch;
    double2 *dpp[2];
    OFX_TRY(d.in(v1, &dv1, n));
    OFX_TRY(d.in(v2, &dv2, n));
    OFX_TRY(d.in(chi, &dchi, n));
    OFX_TRY(d.in(g, &dg, n));
    OFX_TRY(ofx_alloc(ctx, 2 * n, &tmp));
    OFX_TRY(rof_state_in(ctx, p11, p12, &dpp[0], tmp, nx, ny));
    OFX_TRY(rof_state_in(ctx, p21, p22, &dpp[1], tmp, nx, ny));
    OFX_TRY(ofx_alloc(ctx, n, &f1));
    OFX_TRY(ofx_alloc(ctx, n, &f2));
    OFX_TRY(ofx_alloc(ctx, n, &du1));
    OFX_TRY(ofx_alloc(ctx, n, &du2));
    hipLaunchKernelGGL(k_occ_u_init, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)), dim3(64, 4), 0, ctx->stream, (const double *) dv1,
                       (const double *) dv2, (const double *) dchi, f1, f2, du1, du2, nx, ny, theta, beta, OCC_ONE);
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(ofx_alloc(ctx, ROF_SCRATCH_PLANES(2) * rof_skew_elems(nx, ny), &scratch));
    double *us[2] = {du1, du2};
    const double *fs[2] = {f1, f2};
    OFX_TRY(rof_box_dev(ctx, 2, us, fs, dpp, dg, theta, OCC_OMEGA, nx, ny, n_iter, scratch));
    OFX_TRY(d.out(du1, u1, n));
    OFX_TRY(d.out(du2, u2, n));
    OFX_TRY(rof_state_out(ctx, dpp[0], p11, p12, tmp, nx, ny));
    OFX_TRY(rof_state_out(ctx, dpp[1], p21, p22, tmp, nx, ny));
    return d.sync();
}

// ==== TV-L1 with occlusions, the whole solve (src/tvl1occflow.cpp:144-329 single scale, :337-481 multiscale) ===================
// Device-resident from the four uploaded images to the three downloaded planes.  The reference keeps the dual planes of
// Solver_wrt_u and the dual variable of Solver_wrt_chi in function-local statics that it re-creates -- with operator
// new[], uninitialised -- whenever the image width changes, i.e. once per pyramid level; its results are therefore only
// defined on a heap that hands out zeros.  That is the semantics implemented here (state zeroed per level) and pinned by
// the oracle against the reference built with a zero-filling operator new[] (oracle/ref_shim.cpp).
// G independent triples are solved in lockstep (every array is [G][plane], every launch of the outer iteration serves all
// triples that are still iterating, blockIdx.z = triple): one solve is a latency chain of ROF sweeps that keeps ny / 125
// workgroups busy, so G of them cost hardly more time than one.
#define OCC_EXT_MAX_ITERATIONS 20  // src/tvl1occflow_constants.h:35
#define OCC_MAX_ITERATIONS_CHI 100 // :37
#define OCC_MAX_ITERATIONS_U 10    // :36
#define OCC_G_FACTOR 0.05          // :30
#define OCC_TAU_ETA 0.15           // :26
#define OCC_TAU_CHI 0.15           // :27
#define OCC_PRESMOOTHING_SIGMA 0.8 // :34
#define OCC_ERR_BLOCKS 256
#define OCC_MAX_GROUP 16           // triples per lockstep group

// g = 1 / (1 + G_FACTOR |grad filtI0|), choosed_g choice 2 (:96-133); Ix, Iy = centred gradient
__global__ void k_occ_g(const double *__restrict__ Ix, const double *__restrict__ Iy, double *__restrict__ g, size_t size)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    const double gggrad = sqrt(Ix[i] * Ix[i] + Iy[i] * Iy[i]);
    g[i] = 1. / (1. + OCC_G_FACTOR * gggrad);
}

// the six warps of one warping step and what is derived from them (:214-248), one thread per pixel: I1, I1x, I1y sampled at
// x + u, I_1, I_1x, I_1y at x - u (border_out = false), grad = |warped gradient|^2, rho_c = the constant part of the
// linearised residual.  The warped intensities themselves are not used again and stay in registers.
struct OccPrep {
    const double *I0, *I1, *I1x, *I1y, *I_1, *I_1x, *I_1y, *u1, *u2;
    double *I1wx, *I1wy, *I_1wx, *I_1wy, *grad1, *grad3, *rho1_c, *rho3_c;
};
OFX_DEV double occ_sample(const double *__restrict__ in, const BicubicTaps &t, int nx)
{
    double c[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        c[q] = cubic_cell(in[(size_t) t.row[0] * nx + t.col[q]], in[(size_t) t.row[1] * nx + t.col[q]],
                          in[(size_t) t.row[2] * nx + t.col[q]], in[(size_t) t.row[3] * nx + t.col[q]], t.fy);
    return cubic_cell(c[0], c[1], c[2], c[3], t.fx);
}
__global__ void k_occ_prepare(OccPrep a, int nx, int ny, OccGrp G)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    size_t o;
    if (j >= nx || i >= ny || !occ_grp(G, o)) return;
    const size_t p = o + (size_t) i * nx + j;
    const double u1 = a.u1[p], u2 = a.u2[p], i0 = a.I0[p];
    {
        const BicubicTaps t = bicubic_taps(j + u1, i + u2, nx, ny);
        const double w = occ_sample(a.I1 + o, t, nx), wx = occ_sample(a.I1x + o, t, nx), wy = occ_sample(a.I1y + o, t, nx);
        a.I1wx[p] = wx;
        a.I1wy[p] = wy;
        a.grad1[p] = (wx * wx + wy * wy);
        a.rho1_c[p] = (w - wx * u1 - wy * u2 - i0);
    }
    {
        const BicubicTaps t = bicubic_taps(j + (-u1), i + (-u2), nx, ny);
        const double w = occ_sample(a.I_1 + o, t, nx), wx = occ_sample(a.I_1x + o, t, nx), wy = occ_sample(a.I_1y + o, t, nx);
        a.I_1wx[p] = wx;
        a.I_1wy[p] = wy;
        a.grad3[p] = (wx * wx + wy * wy);
        a.rho3_c[p] = (w + wx * u1 + wy * u2 - i0);
    }
}

// 3 x 3 median of both flow components of every triple in one launch (blockIdx.z = 2 triple + component)
__global__ void k_median3_pair(const double *__restrict__ in0, const double *__restrict__ in1, double *__restrict__ out0,
                               double *__restrict__ out1, int nx, int ny, size_t stride, unsigned mask)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, set = blockIdx.z >> 1;
    if (x >= nx || y >= ny || !((mask >> set) & 1u)) return;
    const double *in = ((blockIdx.z & 1) ? in1 : in0) + set * stride;
    double *out = ((blockIdx.z & 1) ? out1 : out0) + set * stride;
    double win[9];
    int n = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            int x0 = x + dx, y0 = y + dy;
            if (x0 < 0) x0 = -x0 - 1;
            if (x0 >= nx) x0 = 2 * nx - x0 - 1;
            if (y0 < 0) y0 = -y0 - 1;
            if (y0 >= ny) y0 = 2 * ny - y0 - 1;
            win[n++] = in[(size_t) y0 * nx + x0];
        }
#pragma unroll
    for (int a = 0; a <= 4; a++) {
#pragma unroll
        for (int b = a + 1; b < 9; b++) {
            const double lo = fmin(win[a], win[b]), hi = fmax(win[a], win[b]);
            win[a] = lo;
            win[b] = hi;
        }
    }
    out[(size_t) y * nx + x] = win[4];
}
// the medians back into u1 / u2 (triples that are still iterating only)
__global__ void k_occ_copy2(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ oa, double *__restrict__ ob,
                            int size, OccGrp G)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    size_t o;
    if (i >= size || !occ_grp(G, o)) return;
    oa[o + i] = a[o + i];
    ob[o + i] = b[o + i];
}

// L2error (:62-80): sum of the squared change of (u1, u2), fixed summation tree (block partials, then one block per triple);
// the previous-iterate planes are refreshed in the same pass
__global__ __launch_bounds__(256) void k_occ_err_partial(const double *__restrict__ u1, const double *__restrict__ u2,
                                                         double *__restrict__ u1p, double *__restrict__ u2p, int size,
                                                         double *__restrict__ part, OccGrp G)
{
    __shared__ double sh[256];
    size_t o;
    if (!occ_grp(G, o)) return;                          // uniform over the workgroup
    double acc = 0.0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < (size_t) size; i += (size_t) OCC_ERR_BLOCKS * 256) {
        const double a = u1[o + i], b = u2[o + i], d1 = a - u1p[o + i], d2 = b - u2p[o + i];
        acc += d1 * d1 + d2 * d2;
        u1p[o + i] = a;
        u2p[o + i] = b;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.z * OCC_ERR_BLOCKS + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_occ_err_final(const double *__restrict__ part, double *__restrict__ out)
{
    __shared__ double sh[256];
    sh[threadIdx.x] = threadIdx.x < OCC_ERR_BLOCKS ? part[blockIdx.x * OCC_ERR_BLOCKS + threadIdx.x] : 0.0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
__global__ void k_occ_scale2(double *__restrict__ a, double *__restrict__ b, size_t n, double s)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] *= s;
        b[i] *= s;
    }
}
__global__ void k_occ_threshold(double *__restrict__ chi, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) chi[i] = (chi[i] > OCC_THR_CHI);
}

namespace {

struct OccParams {
    double lambda, alpha, beta, theta, epsilon;
    int warps, verbose;
};
// work planes of a group of G solves, allocated once at the finest level's size; a level uses them as [G][nx ny] /
// [G][rof_skew_elems()]
struct OccWork {
    double *I1x, *I1y, *I_1x, *I_1y, *I1wx, *I1wy, *I_1wx, *I_1wy, *rho1_c, *rho3_c, *grad1, *grad3, *v1, *v2, *vf1, *vf2, *vb1,
        *vb2, *g, *u1p, *u2p, *f1, *f2, *t1, *t2, *divu, *state, *eta, *rof, *part, *chi_alt;
    int alloc(ofx_ctx *ctx, size_t n, size_t nskew, int G)
    {
        double **planes[] = {&I1x, &I1y, &I_1x, &I_1y, &I1wx, &I1wy, &I_1wx, &I_1wy, &rho1_c, &rho3_c, &grad1, &grad3, &v1,
                             &v2, &vf1, &vf2, &vb1, &vb2, &g, &u1p, &u2p, &f1, &f2, &t1, &t2, &divu};
        for (auto p : planes) OFX_TRY(ofx_alloc(ctx, n * G, p));
        OFX_TRY(ofx_alloc(ctx, 4 * nskew * G, &state));      // dual planes of Solver_wrt_u, hyperplane-major
        OFX_TRY(ofx_alloc(ctx, 2 * n * G, &eta));            // dual variable of Solver_wrt_chi
        OFX_TRY(ofx_alloc(ctx, 3 * n * G, &chi_alt));        // second set of (chi, eta1, eta2) planes of the fused chi iterations
        OFX_TRY(ofx_alloc(ctx, ROF_SCRATCH_PLANES(2) * nskew * G, &rof));
        return ofx_alloc(ctx, (size_t) (OCC_ERR_BLOCKS + 1) * G, &part);
    }
};

// one level of G triples, arrays [G][nx ny] on the device, u1 / u2 / chi in place; the level's dual state is zeroed here.
// stats[g]: the record of triple g.
int occ_single_scale_dev(ofx_ctx *ctx, int G, const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1,
                         double *u2, double *chi, int nx, int ny, const OccParams &P, const OccWork &W, int scale, ofx_stats *stats)
{
    const size_t n = (size_t) nx * ny;
    const dim3 grid(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G), block(64, 4);
    const dim3 g1(occ_grid1d(n), 1, G);
    hipStream_t st = ctx->stream;
    const size_t nsk = rof_skew_elems(nx, ny);
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    OFX_HIP(ctx, hipMemsetAsync(W.state, 0, 4 * nsk * G * sizeof(double), st));
    OFX_HIP(ctx, hipMemsetAsync(W.eta, 0, 2 * n * G * sizeof(double), st));
    double *eta1 = W.eta, *eta2 = W.eta + n * G;
    for (int g = 0; g < G; g++) {                         // once per level: plane by plane
        const size_t o = g * n;
        OFX_TRY(op_centered_gradient<double>(ctx, filtI0 + o, W.t1 + o, W.t2 + o, nx, ny));
        OFX_TRY(op_centered_gradient<double>(ctx, I1 + o, W.I1x + o, W.I1y + o, nx, ny));
        OFX_TRY(op_centered_gradient<double>(ctx, I_1 + o, W.I_1x + o, W.I_1y + o, nx, ny));
    }
    hipLaunchKernelGGL(k_occ_g, dim3(occ_grid1d(n * G)), dim3(256), 0, st, (const double *) W.t1, (const double *) W.t2, W.g, n * G);
    OFX_HIP(ctx, hipMemcpyAsync(W.u1p, u1, n * G * sizeof(double), hipMemcpyDeviceToDevice, st));
    OFX_HIP(ctx, hipMemcpyAsync(W.u2p, u2, n * G * sizeof(double), hipMemcpyDeviceToDevice, st));
    const OccPrep prep = {I0, I1, W.I1x, W.I1y, I_1, W.I_1x, W.I_1y, u1, u2, W.I1wx, W.I1wy, W.I_1wx, W.I_1wy, W.grad1, W.grad3,
                          W.rho1_c, W.rho3_c};
    const OccV av = {u1, u2, chi, W.I1wx, W.I1wy, W.I_1wx, W.I_1wy, W.rho1_c, W.rho3_c, W.grad1, W.grad3,
                     W.v1, W.v2, W.vf1, W.vf2, W.vb1, W.vb2};
    const OccChi ac = {u1, u2, W.I1wx, W.I1wy, W.I_1wx, W.I_1wy, W.rho1_c, W.rho3_c, W.vf1, W.vf2, W.vb1, W.vb2, W.g, eta1, eta2,
                       W.divu, chi};
    double *us[2] = {u1, u2};
    double2 *pps[2] = {reinterpret_cast<double2 *>(W.state), reinterpret_cast<double2 *>(W.state + 2 * nsk * G)};
    const double *fs[2] = {W.f1, W.f2};
    double *h_err = ctx->h_aux;                                      // pinned, OFX_MAX_GROUP doubles, not the poll ring
    double *d_err = W.part + (size_t) OCC_ERR_BLOCKS * G;
    for (int w = 0; w < P.warps; w++) {
        hipLaunchKernelGGL(k_occ_prepare, grid, block, 0, st, prep, nx, ny, OccGrp{n, all});
        OFX_LAUNCH_CHECK(ctx);
        int it[OCC_MAX_GROUP];
        double error[OCC_MAX_GROUP];
        for (int g = 0; g < G; g++) { it[g] = 0; error[g] = INFINITY; }
        unsigned active = all;
        while (active) {                                  // while (error > epsilon && it < EXT_MAX_ITERATIONS), per triple
            const OccGrp grp = {n, active};
            hipLaunchKernelGGL(k_occ_v, g1, dim3(256), 0, st, av, (int) n, P.alpha, P.theta, P.lambda, grp);
            hipLaunchKernelGGL(k_occ_u_init, grid, block, 0, st, (const double *) W.v1, (const double *) W.v2, (const double *) chi,
                               W.f1, W.f2, u1, u2, nx, ny, P.theta, P.beta, grp);
            OFX_LAUNCH_CHECK(ctx);
            OFX_TRY(rof_box_dev(ctx, 2, us, fs, pps, W.g, P.theta, OCC_OMEGA, nx, ny, OCC_MAX_ITERATIONS_U, W.rof, G, active));
            hipLaunchKernelGGL(k_median3_pair, dim3(grid.x, grid.y, 2 * G), block, 0, st, (const double *) u1, (const double *) u2, W.t1,
                               W.t2, nx, ny, n, active);
            hipLaunchKernelGGL(k_occ_copy2, g1, dim3(256), 0, st, (const double *) W.t1, (const double *) W.t2, u1, u2, (int) n, grp);
            hipLaunchKernelGGL(k_occ_divu, grid, block, 0, st, (const double *) u1, (const double *) u2, W.divu, nx, ny, grp);
            OFX_TRY(occ_chi_iterations(ctx, ac, chi, eta1, eta2, W.chi_alt, nx, ny, OCC_MAX_ITERATIONS_CHI, P.lambda, P.theta, P.alpha,
                                       P.beta, OCC_TAU_CHI, OCC_TAU_ETA, G, grp));
            hipLaunchKernelGGL(k_occ_err_partial, dim3(OCC_ERR_BLOCKS, 1, G), dim3(256), 0, st, (const double *) u1, (const double *) u2,
                               W.u1p, W.u2p, (int) n, W.part, grp);
            hipLaunchKernelGGL(k_occ_err_final, dim3(G), dim3(256), 0, st, (const double *) W.part, d_err);
            OFX_LAUNCH_CHECK(ctx);
            OFX_HIP(ctx, hipMemcpyAsync(h_err, d_err, G * sizeof(double), hipMemcpyDeviceToHost, st));
            OFX_HIP(ctx, hipStreamSynchronize(st));
            for (int g = 0; g < G; g++) {
                if (!((active >> g) & 1u)) continue;
                it[g]++;
                error[g] = h_err[g] / (int) n;
                if (!(error[g] > P.epsilon && it[g] < OCC_EXT_MAX_ITERATIONS)) active &= ~(1u << g);
            }
        }
        if (P.verbose && G == 1) fprintf(stderr, "Warping: %d, Iterations: %d, Error: %e\n", w, it[0], error[0]);     // :292-296
        for (int g = 0; g < G; g++) {
            ofx_stats &S = stats[g];
            if (scale < OFX_MAX_SCALES && w < OFX_MAX_SOLVES) {
                S.iters[scale][w] = it[g];
                S.error[scale][w] = error[g];
            }
            S.work_pix_iters += (double) it[g] * (double) n;
        }
    }
    return OFX_OK;
}

// G triples (host planes) on one context
int occ_group_impl(ofx_ctx *ctx, int G, const double *const *I_1, const double *const *I0, const double *const *I1,
                   const double *const *filtI0, double *const *u1, double *const *u2, double *const *chi, int nxx, int nyy,
                   const OccParams &P, int nscales, double zfactor, ofx_stats *stats);
// An error return must not leave kernels or copies into the caller's planes in flight (the next call resets the arena they
// work in): drain the stream first, as ofx_run_loop_group does for the iteration loops.
int occ_group(ofx_ctx *ctx, int G, const double *const *I_1, const double *const *I0, const double *const *I1,
              const double *const *filtI0, double *const *u1, double *const *u2, double *const *chi, int nxx, int nyy,
              const OccParams &P, int nscales, double zfactor, ofx_stats *stats)
{
    const int s = occ_group_impl(ctx, G, I_1, I0, I1, filtI0, u1, u2, chi, nxx, nyy, P, nscales, zfactor, stats);
    if (s != OFX_OK) (void) hipStreamSynchronize(ctx->stream);
    return s;
}
int occ_group_impl(ofx_ctx *ctx, int G, const double *const *I_1, const double *const *I0, const double *const *I1,
                   const double *const *filtI0, double *const *u1, double *const *u2, double *const *chi, int nxx, int nyy,
                   const OccParams &P, int nscales, double zfactor, ofx_stats *stats)
{
    const double t0 = ofx_now_ms();
    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nxx, nyy, nscales, zfactor, nxs, nys));
    if (nxs[nscales - 1] < 2 || nys[nscales - 1] < 2) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: coarsest level %dx%d", nxs[nscales - 1], nys[nscales - 1]);
    for (int g = 0; g < G; g++) {
        ofx_stats &S = stats[g];
        S = ofx_stats{};
        S.nscales = nscales;
        S.nsolves = P.warps;
        for (int s = 0; s < nscales && s < OFX_MAX_SCALES; s++) { S.nx[s] = nxs[s]; S.ny[s] = nys[s]; }
    }
    const size_t size = (size_t) nxx * nyy;
    OccWork W;
    OFX_TRY(W.alloc(ctx, size, rof_skew_elems(nxx, nyy), G));
    struct Lv { double *im[4], *u1, *u2, *chi; };           // im: I_1, I0, I1, filtI0; every array [G][n_s]
    std::vector<Lv> lv(nscales);
    for (int s = 0; s < nscales; s++) {
        const size_t n = (size_t) nxs[s] * nys[s] * G;
        for (int k = 0; k < 4; k++) OFX_TRY(ofx_alloc(ctx, n, &lv[s].im[k]));
        OFX_TRY(ofx_alloc(ctx, n, &lv[s].u1));
        OFX_TRY(ofx_alloc(ctx, n, &lv[s].u2));
        OFX_TRY(ofx_alloc(ctx, n, &lv[s].chi));
    }
    hipStream_t st = ctx->stream;
    // :382-395 call image_normalization_4 and then overwrite its output with the raw images: no normalisation
    const double *const *host[4] = {I_1, I0, I1, filtI0};
    for (int g = 0; g < G; g++)
        for (int k = 0; k < 4; k++) {
            OFX_HIP(ctx, hipMemcpyAsync(lv[0].im[k] + g * size, host[k][g], size * sizeof(double), hipMemcpyHostToDevice, st));
            OFX_TRY(op_gaussian<double>(ctx, lv[0].im[k] + g * size, W.t1, nxx, nyy, OCC_PRESMOOTHING_SIGMA));
        }
    for (int s = 1; s < nscales; s++) {
        const size_t np = (size_t) nxs[s - 1] * nys[s - 1], n = (size_t) nxs[s] * nys[s];
        for (int g = 0; g < G; g++)
            for (int k = 0; k < 4; k++)
                OFX_TRY(op_zoom_out<double>(ctx, lv[s - 1].im[k] + g * np, lv[s].im[k] + g * n, W.t1, W.t2, nxs[s - 1], nys[s - 1], zfactor));
    }
    {
        const size_t n = (size_t) nxs[nscales - 1] * nys[nscales - 1] * G;
        OFX_HIP(ctx, hipMemsetAsync(lv[nscales - 1].u1, 0, n * sizeof(double), st));
        OFX_HIP(ctx, hipMemsetAsync(lv[nscales - 1].u2, 0, n * sizeof(double), st));
        OFX_HIP(ctx, hipMemsetAsync(lv[nscales - 1].chi, 0, n * sizeof(double), st));
    }
    for (int s = nscales - 1; s >= 0; s--) {
        if (P.verbose && G == 1) fprintf(stderr, "Scale %d: %dx%d\n", s, nxs[s], nys[s]);
        OFX_TRY(occ_single_scale_dev(ctx, G, lv[s].im[0], lv[s].im[1], lv[s].im[2], lv[s].im[3], lv[s].u1, lv[s].u2, lv[s].chi, nxs[s],
                                     nys[s], P, W, s, stats));
        if (s) {
            const double fx = (double) nxs[s - 1] / nxs[s], fy = (double) nys[s - 1] / nys[s];
            const size_t n = (size_t) nxs[s - 1] * nys[s - 1], nc = (size_t) nxs[s] * nys[s];
            for (int g = 0; g < G; g++) {
                OFX_TRY(op_resample<double>(ctx, lv[s].u1 + g * nc, lv[s - 1].u1 + g * n, nxs[s], nys[s], nxs[s - 1], nys[s - 1], fx, fy));
                OFX_TRY(op_resample<double>(ctx, lv[s].u2 + g * nc, lv[s - 1].u2 + g * n, nxs[s], nys[s], nxs[s - 1], nys[s - 1], fx, fy));
                OFX_TRY(op_resample<double>(ctx, lv[s].chi + g * nc, lv[s - 1].chi + g * n, nxs[s], nys[s], nxs[s - 1], nys[s - 1], fx, fy));
            }
            hipLaunchKernelGGL(k_occ_scale2, dim3(occ_grid1d(n * G)), dim3(256), 0, st, lv[s - 1].u1, lv[s - 1].u2, n * G, (double) 1.0 / zfactor);
        } else {
            hipLaunchKernelGGL(k_occ_threshold, dim3(occ_grid1d(size * G)), dim3(256), 0, st, lv[0].chi, size * G);
        }
        OFX_LAUNCH_CHECK(ctx);
    }
    Dev d{ctx};
    for (int g = 0; g < G; g++) {
        OFX_TRY(d.out(lv[0].u1 + g * size, u1[g], size));
        OFX_TRY(d.out(lv[0].u2 + g * size, u2[g], size));
        OFX_TRY(d.out(lv[0].chi + g * size, chi[g], size));
    }
    OFX_TRY(d.sync());
    const double ms = ofx_now_ms() - t0;
    for (int g = 0; g < G; g++) stats[g].total_ms = ms;
    return OFX_OK;
}

int occ_check_args(ofx_ctx *ctx, int nxx, int nyy, double lambda, double theta, int nscales, double zfactor, int warps)
{
    if (nxx < 2 || nyy < 2 || (long long) nxx * nyy > 0x3fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: bad size %dx%d", nxx, nyy);
    if (nscales < 1 || warps < 1 || warps > OFX_MAX_SOLVES) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: nscales=%d warps=%d", nscales, warps);
    if (!(zfactor > 0.0 && zfactor < 1.0)) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: zfactor=%g", zfactor);
    if (!(theta > 0.0) || !(lambda > 0.0)) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: theta=%g lambda=%g", theta, lambda);
    return OFX_OK;
}

}   // namespace

extern "C" int ofx_tvl1occ_multiscale(ofx_ctx *ctx, const double *I_1, const double *I0, const double *I1, const double *filtI0,
                                      double *u1, double *u2, double *chi, int nxx, int nyy, double lambda, double alpha,
                                      double beta, double theta, int nscales, double zfactor, int warps, double epsilon,
                                      int verbose)
{
    OFX_ENTER(ctx);
    if (!I_1 || !I0 || !I1 || !filtI0 || !u1 || !u2 || !chi) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1occ: NULL pointer");
    OFX_TRY(occ_check_args(ctx, nxx, nyy, lambda, theta, nscales, zfactor, warps));
    const OccParams P = {lambda, alpha, beta, theta, epsilon, warps, verbose};
    return occ_group(ctx, 1, &I_1, &I0, &I1, &filtI0, &u1, &u2, &chi, nxx, nyy, P, nscales, zfactor, &ctx->stats);
}

// Several independent triples (e.g. the frames of a sequence): cut into lockstep groups of up to 16 consecutive triples (fewer
// when the device memory the contexts may use -- option "mem_budget" -- does not hold that many), group q on context
// q mod n_ctx, one host thread per context; all contexts on one device.
extern "C" int ofx_tvl1occ_batch(ofx_ctx *const *ctxs, int n_ctx, int n_triples, const double *const *I_1, const double *const *I0,
                                 const double *const *I1, const double *const *filtI0, double *const *u1, double *const *u2,
                                 double *const *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta,
                                 int nscales, double zfactor, int warps, double epsilon)
{
    if (!ctxs || n_ctx < 1 || n_triples < 0 || !I_1 || !I0 || !I1 || !filtI0 || !u1 || !u2 || !chi) return OFX_ERR_ARG;
    for (int w = 0; w < n_ctx; w++)
        if (!ctxs[w] || ctxs[w]->device != ctxs[0]->device) return OFX_ERR_ARG;
    if (n_triples == 0) return OFX_OK;
    {
        ofx_ctx *ctx = ctxs[0];
        OFX_ENTER(ctx);
        OFX_TRY(occ_check_args(ctx, nxx, nyy, lambda, theta, nscales, zfactor, warps));
    }
    // group size: option "lockstep" of ctxs[0], else as many as fit.  Per triple: 7 planes per pyramid level (occ_group: 4 images,
    // u1, u2, chi) + OccWork's 31 full-size row-major and 13 hyperplane-major planes + 2 of Gaussian / zoom scratch
    int G = ctxs[0]->lockstep > 0 ? ctxs[0]->lockstep : OCC_MAX_GROUP;
    if (G > OCC_MAX_GROUP) G = OCC_MAX_GROUP;
    {
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(ctxs[0]->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return OFX_ERR_HIP;
        const double budget = (ctxs[0]->mem_budget > 0 ? ctxs[0]->mem_budget : 0.5 * (double) free_b) / n_ctx;
        std::vector<int> nxs, nys;
        OFX_TRY(op_pyramid_sizes(ctxs[0], nxx, nyy, nscales, zfactor, nxs, nys));
        double level_px = 0.0;
        for (int s = 0; s < nscales; s++) level_px += (double) nxs[s] * nys[s];
        const double per = 8.0 * (7.0 * level_px + 33.0 * nxx * nyy + 13.0 * (double) rof_skew_elems(nxx, nyy)) * 1.05;
        const int fit = (int) (budget / per);
        if (fit < 1) return ofx_fail(ctxs[0], OFX_ERR_NOMEM, "tvl1occ batch: %.1f GB per triple, %.1f GB per context available", per / 1e9, budget / 1e9);
        if (G > fit) G = fit;
    }
    const int per_ctx = (n_triples + n_ctx - 1) / n_ctx;     // even the groups out over the contexts
    if (G > per_ctx) G = per_ctx;
    const int n_groups = (n_triples + G - 1) / G;
    const OccParams P = {lambda, alpha, beta, theta, epsilon, warps, 0};
    std::atomic<int> status(OFX_OK);
    auto worker = [&](int w) {
        std::vector<ofx_stats> st(G);
        for (int q = w; q < n_groups; q += n_ctx) {
            if (status.load() != OFX_OK) return;
            const int first = q * G, cnt = (n_triples - first < G) ? n_triples - first : G;
            ofx_ctx *ctx = ctxs[w];
            int s = OFX_OK;
            if (hipSetDevice(ctx->device) != hipSuccess) s = ofx_fail(ctx, OFX_ERR_NODEV, "hipSetDevice(%d) failed", ctx->device);
            else {
                ctx->errmsg[0] = 0;
                ofx_arena_reset(ctx);
                s = occ_group(ctx, cnt, I_1 + first, I0 + first, I1 + first, filtI0 + first, u1 + first, u2 + first, chi + first, nxx,
                              nyy, P, nscales, zfactor, st.data());
                ctx->stats = st[0];
            }
            if (s != OFX_OK) { int expected = OFX_OK; status.compare_exchange_strong(expected, s); return; }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_ctx && w < n_groups; w++) th.emplace_back(worker, w);
    worker(0);
    for (auto &t : th) t.join();
    return status.load();
}
