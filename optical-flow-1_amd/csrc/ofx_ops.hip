// ofx_ops.hip -- gfx950 kernels + launchers for the shared operators and the pyramid
// (reference: src/operators.cpp, src/bicubic_interpolation.cpp, src/zoom.cpp, src/utils.cpp).
//
// All of these are HBM-bound elementwise / small-stencil / gather kernels: one pixel per lane,
// x fastest so every wave touches 64 consecutive pixels of a row (coalesced), 64x4 blocks so a
// block's four waves share their vertical neighbours through L1.  They run once per pyramid level
// or once per warp, never inside the inner iteration.
#include "ofx_ops.h"
#include "ofx_device.h"

#include <cmath>

#define BX 64
#define BY 4

static inline dim3 grid2d(int nx, int ny) { return dim3(ofx_cdiv(nx, BX), ofx_cdiv(ny, BY)); }
static inline dim3 block2d() { return dim3(BX, BY); }
static inline int grid1d(size_t n) { return (int) ((n + 255) / 256); }

// ---- layout conversions ---------------------------------------------------------------------------
template <typename T>
__global__ void k_convert_in(const double *__restrict__ src, T *__restrict__ dst, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stn(dst + i, src[i]);
}

template <typename T>
__global__ void k_convert_out(const T *__restrict__ src, double *__restrict__ dst, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ldw(src + i);
}

template <typename T>
__global__ void k_interleave2(const double *__restrict__ a, const double *__restrict__ b,
                              typename Pix<T>::v2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stn2(dst + i, make_double2(a[i], b[i]));
}

template <typename T>
__global__ void k_deinterleave2(const typename Pix<T>::v2 *__restrict__ src, double *__restrict__ a,
                                double *__restrict__ b, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double2 v = ldw2(src + i);
        a[i] = v.x;
        b[i] = v.y;
    }
}

// (u,v) -> float32 pairs: the cast of src/tvl1flow_main.cpp:209-213
template <typename T>
__global__ void k_to_flo(const typename Pix<T>::v2 *__restrict__ src, float2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double2 v = ldw2(src + i);
        dst[i] = make_float2((float) v.x, (float) v.y);
    }
}

template <typename T> int op_convert_in(ofx_ctx *ctx, const double *src, T *dst, size_t n)
{
    hipLaunchKernelGGL(k_convert_in<T>, dim3(grid1d(n)), dim3(256), 0, ctx->stream, src, dst, n);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_convert_out(ofx_ctx *ctx, const T *src, double *dst, size_t n)
{
    hipLaunchKernelGGL(k_convert_out<T>, dim3(grid1d(n)), dim3(256), 0, ctx->stream, src, dst, n);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T>
int op_interleave2(ofx_ctx *ctx, const double *a, const double *b, typename Pix<T>::v2 *dst, size_t n)
{
    hipLaunchKernelGGL(k_interleave2<T>, dim3(grid1d(n)), dim3(256), 0, ctx->stream, a, b, dst, n);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T>
int op_deinterleave2(ofx_ctx *ctx, const typename Pix<T>::v2 *src, double *a, double *b, size_t n)
{
    hipLaunchKernelGGL(k_deinterleave2<T>, dim3(grid1d(n)), dim3(256), 0, ctx->stream, src, a, b, n);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_to_flo(ofx_ctx *ctx, const typename Pix<T>::v2 *src, float2 *dst, size_t n)
{
    hipLaunchKernelGGL(k_to_flo<T>, dim3(grid1d(n)), dim3(256), 0, ctx->stream, src, dst, n);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_fill2(ofx_ctx *ctx, typename Pix<T>::v2 *dst, size_t n)
{
    OFX_HIP(ctx, hipMemsetAsync(dst, 0, n * sizeof(typename Pix<T>::v2), ctx->stream));
    return OFX_OK;
}

// ---- image_normalization_2 (src/utils.cpp:283-326, getminmax :509-525) -----------------------------
// min/max are exact (order-independent); two-stage reduction: per-block partials, then one block.
#define MM_BLOCKS 1024
#define MM_SCR (2 * MM_BLOCKS + 2)      // doubles of scratch per (pair of) image(s): partial minima, partial maxima, {min, max}
template <typename T>
OFX_DEV void minmax_partial_block(const T *__restrict__ I1, const T *__restrict__ I2, int size, double *__restrict__ part)
{
    double lo = ldw(I1), hi = lo;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < (size_t) size;
         i += (size_t) gridDim.x * blockDim.x) {
        const double a = ldw(I1 + i), b = ldw(I2 + i);
        lo = a < lo ? a : lo; hi = a > hi ? a : hi;
        lo = b < lo ? b : lo; hi = b > hi ? b : hi;
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
        part[MM_BLOCKS + blockIdx.x] = hi;
    }
}
OFX_DEV void minmax_final_block(const double *__restrict__ part, int nblocks, double *__restrict__ mm)
{
    double lo = part[0], hi = part[MM_BLOCKS];
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        lo = part[i] < lo ? part[i] : lo;
        hi = part[MM_BLOCKS + i] > hi ? part[MM_BLOCKS + i] : hi;
    }
    lo = wave_allreduce_min(lo);
    hi = wave_allreduce_max(hi);
    __shared__ double slo[4], shi[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[w] = lo; shi[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { lo = slo[k] < lo ? slo[k] : lo; hi = shi[k] > hi ? shi[k] : hi; }
        mm[0] = lo;
        mm[1] = hi;
    }
}
template <typename T>
OFX_DEV void normalize2_px(const T *__restrict__ I1, const T *__restrict__ I2, T *__restrict__ o1, T *__restrict__ o2, int size,
                           const double *__restrict__ mm)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) size) return;
    const double lo = mm[0], den = mm[1] - mm[0];
    const double a = ldw(I1 + i), b = ldw(I2 + i);
    if (den > 0) {
        stn(o1 + i, 255.0 * (a - lo) / den);
        stn(o2 + i, 255.0 * (b - lo) / den);
    } else {
        stn(o1 + i, a);
        stn(o2 + i, b);
    }
}

template <typename T>
__global__ void k_minmax_partial(const T *__restrict__ I1, const T *__restrict__ I2, int size,
                                 double *__restrict__ part /* [2][MM_BLOCKS] */)
{
    minmax_partial_block<T>(I1, I2, size, part);
}
__global__ void k_minmax_final(const double *__restrict__ part, int nblocks, double *__restrict__ mm)
{
    minmax_final_block(part, nblocks, mm);
}
template <typename T>
__global__ void k_normalize2(const T *__restrict__ I1, const T *__restrict__ I2, T *__restrict__ o1,
                             T *__restrict__ o2, int size, const double *__restrict__ mm)
{
    normalize2_px<T>(I1, I2, o1, o2, size, mm);
}
// the same three kernels for the G pairs of a lockstep group in one launch each (blockIdx.y = pair); the inputs are the
// caller's separate device images, the outputs level arrays that hold the pairs back to back
template <typename T> __global__ void k_minmax_partial_g(OfxGroupPtrs P, int size, double *__restrict__ scr)
{
    const int g = blockIdx.y;
    minmax_partial_block<T>((const T *) P.a[g], (const T *) P.b[g], size, scr + (size_t) g * MM_SCR);
}
__global__ void k_minmax_final_g(double *__restrict__ scr, int nblocks)
{
    double *s = scr + (size_t) blockIdx.x * MM_SCR;
    minmax_final_block(s, nblocks, s + 2 * MM_BLOCKS);
}
template <typename T>
__global__ void k_normalize2_g(OfxGroupPtrs P, T *__restrict__ o1, T *__restrict__ o2, int size, const double *__restrict__ scr)
{
    const int g = blockIdx.y;
    normalize2_px<T>((const T *) P.a[g], (const T *) P.b[g], o1 + (size_t) g * size, o2 + (size_t) g * size, size,
                     scr + (size_t) g * MM_SCR + 2 * MM_BLOCKS);
}

template <typename T>
int op_normalize2(ofx_ctx *ctx, const T *I1, const T *I2, T *o1, T *o2, int size, double *scr)
{
    int nb = grid1d((size_t) size);
    if (nb > MM_BLOCKS) nb = MM_BLOCKS;
    double *part = scr, *mm = scr + 2 * MM_BLOCKS;
    hipLaunchKernelGGL(k_minmax_partial<T>, dim3(nb), dim3(256), 0, ctx->stream, I1, I2, size, part);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, ctx->stream, part, nb, mm);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_normalize2<T>, dim3(grid1d((size_t) size)), dim3(256), 0, ctx->stream, I1, I2, o1, o2,
                       size, mm);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// ---- gaussian (src/operators.cpp:506-624) -----------------------------------------------------------
int ofx_gauss_taps(double sigma, GaussTaps *t)
{
    const double den = 2 * sigma * sigma;
    const int size = (int) (5 * sigma) + 1;                 // DEFAULT_GAUSSIAN_WINDOW_SIZE, operators.h:120
    if (size < 1 || size > OFX_GAUSS_MAX_TAPS) return OFX_ERR_ARG;
    t->size = size;
    t->dirichlet = 0;
    for (int i = 0; i < size; i++)
        t->B[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);   // :527 (pi truncated)
    double norm = 0;
    for (int i = 0; i < size; i++) norm += t->B[i];
    norm *= 2;
    norm -= t->B[0];
    for (int i = 0; i < size; i++) t->B[i] /= norm;
    return OFX_OK;
}

// reflecting boundary of :557-562: left of the image the edge sample is NOT repeated (t=-1 -> 1),
// right of it it IS repeated (t=n -> n-1).
OFX_DEV int gauss_reflect(int t, int n) { return t < 0 ? -t : (t >= n ? 2 * n - 1 - t : t); }

// Planes of a batched launch: blockIdx.z in [0, 2 G) = image (A, B) x pair; the G planes of an image are `stride` apart
template <typename T> struct OfxPlanes2 {
    T     *a, *b;
    int    G;
    size_t stride;
    OFX_DEV T *at(int z) const { return (z < G ? a : b) + (size_t) (z < G ? z : z - G) * stride; }
};
template <typename T, bool ALONG_X>
OFX_DEV void gauss_pass_px(const T *__restrict__ in, T *__restrict__ out, int nx, int ny, const GaussTaps &taps)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    double sum = taps.B[0] * ldw(in + p);
    if (taps.dirichlet) {                                    // BOUNDARY_CONDITION_DIRICHLET (:551-555, :591-595): 0 outside the image
        if (ALONG_X) {
            const T *row = in + (size_t) i * nx;
            for (int k = 1; k < taps.size; k++)
                sum += taps.B[k] * ((j - k >= 0 ? ldw(row + j - k) : 0.0) + (j + k < nx ? ldw(row + j + k) : 0.0));
        } else {
            for (int k = 1; k < taps.size; k++)
                sum += taps.B[k] * ((i - k >= 0 ? ldw(in + (size_t) (i - k) * nx + j) : 0.0) +
                                    (i + k < ny ? ldw(in + (size_t) (i + k) * nx + j) : 0.0));
        }
        stn(out + p, sum);
        return;
    }
    if (ALONG_X) {
        const T *row = in + (size_t) i * nx;
        for (int k = 1; k < taps.size; k++)
            sum += taps.B[k] * (ldw(row + gauss_reflect(j - k, nx)) + ldw(row + gauss_reflect(j + k, nx)));
    } else {
        for (int k = 1; k < taps.size; k++)
            sum += taps.B[k] * (ldw(in + (size_t) gauss_reflect(i - k, ny) * nx + j) +
                                ldw(in + (size_t) gauss_reflect(i + k, ny) * nx + j));
    }
    stn(out + p, sum);
}
template <typename T, bool ALONG_X>
__global__ void k_gauss_pass(const T *__restrict__ in, T *__restrict__ out, int nx, int ny, GaussTaps taps)
{
    gauss_pass_px<T, ALONG_X>(in, out, nx, ny, taps);
}
template <typename T, bool ALONG_X>
__global__ void k_gauss_pass_g(OfxPlanes2<const T> in, OfxPlanes2<T> out, int nx, int ny, GaussTaps taps)
{
    gauss_pass_px<T, ALONG_X>(in.at(blockIdx.z), out.at(blockIdx.z), nx, ny, taps);
}

// Both passes in one launch for the pyramids of a lockstep group: a block brings a (GXY_TH + 2 r) x (64 + 2 r) region of the
// input into LDS (reflected indices resolved while loading), runs the row pass on all of its rows, then the column pass on the
// GXY_TH inner ones -- the intermediate image never touches memory (the two-pass form writes and re-reads it: 4 plane moves
// instead of 2).  Per pixel the same sums in the same order as k_gauss_pass, the intermediate rounded to the storage type as
// the two-pass form stores it.  in != out (neighbouring blocks read what a block would overwrite).  Radius <= GXY_RMAX.
#define GXY_TH 32
#define GXY_RMAX 8
template <typename T>
__global__ __launch_bounds__(256) void k_gauss_xy_g(OfxPlanes2<const T> in_p, OfxPlanes2<T> out_p, int nx, int ny, GaussTaps taps)
{
    __shared__ double s_in[(GXY_TH + 2 * GXY_RMAX) * (64 + 2 * GXY_RMAX)];
    __shared__ double s_mid[(GXY_TH + 2 * GXY_RMAX) * 64];
    const T *__restrict__ in = in_p.at(blockIdx.z);
    T *__restrict__ out = out_p.at(blockIdx.z);
    const int R = taps.size - 1, W = 64 + 2 * R, H = GXY_TH + 2 * R;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * GXY_TH;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int k = tid; k < W * H; k += 256) {
        const int r = k / W, c = k - r * W;
        // rows / columns beyond the image on the far side of a border block are never used: clamp them into range
        int i = gauss_reflect(i0 - R + r, ny), j = gauss_reflect(j0 - R + c, nx);
        i = i < 0 ? 0 : (i > ny - 1 ? ny - 1 : i);
        j = j < 0 ? 0 : (j > nx - 1 ? nx - 1 : j);
        s_in[r * W + c] = ldw(in + (size_t) i * nx + j);
    }
    __syncthreads();
    for (int r = threadIdx.y; r < H; r += 4) {                   // row pass, :541-575
        const double *row = s_in + r * W + R + threadIdx.x;
        double sum = taps.B[0] * row[0];
        for (int k = 1; k < taps.size; k++) sum += taps.B[k] * (row[-k] + row[k]);
        s_mid[r * 64 + threadIdx.x] = sizeof(T) == sizeof(float) ? (double) (float) sum : sum;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= nx) return;
    for (int r = threadIdx.y; r < GXY_TH; r += 4) {              // column pass, :577-611
        const int i = i0 + r;
        if (i >= ny) break;
        const double *col = s_mid + (r + R) * 64 + threadIdx.x;
        double sum = taps.B[0] * col[0];
        for (int k = 1; k < taps.size; k++) sum += taps.B[k] * (col[-k * 64] + col[k * 64]);
        stn(out + (size_t) i * nx + j, sum);
    }
}

// The same with the radius R a template parameter (the pyramids use 4 and 5).  k_gauss_xy_g above spends its time on index
// arithmetic (a division, two reflections and a 64-bit address per loaded element) and on 2 R + 1 LDS reads per output of both
// passes -- 1.2 TB/s of plane traffic.  Here the reflected column of a thread is computed once (it does not depend on the row),
// the reflected row once per row, and the column pass slides a register window down the thread's column: 8 + 2 R reads for 8
// outputs.  Same sums in the same order per pixel.
template <typename T, int R>
__global__ __launch_bounds__(256) void k_gauss_xy_gr(OfxPlanes2<const T> in_p, OfxPlanes2<T> out_p, int nx, int ny, GaussTaps taps)
{
    constexpr int W = 64 + 2 * R, H = GXY_TH + 2 * R;
    __shared__ double s_in[H * W];
    __shared__ double s_mid[H * 64];
    const T *__restrict__ in = in_p.at(blockIdx.z);
    T *__restrict__ out = out_p.at(blockIdx.z);
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * GXY_TH;
    const int tx = threadIdx.x, ty = threadIdx.y;
    // columns of this thread in the staged region: tx, and tx + 64 for the first 2 R lanes; beyond the image on the far side of
    // a border block: never used, clamped into range
    int ja = gauss_reflect(j0 - R + tx, nx), jb = gauss_reflect(j0 - R + tx + 64, nx);
    ja = ja < 0 ? 0 : (ja > nx - 1 ? nx - 1 : ja);
    jb = jb < 0 ? 0 : (jb > nx - 1 ? nx - 1 : jb);
    for (int r = ty; r < H; r += 4) {
        int i = gauss_reflect(i0 - R + r, ny);
        i = i < 0 ? 0 : (i > ny - 1 ? ny - 1 : i);
        const T *__restrict__ row = in + (size_t) i * nx;
        s_in[r * W + tx] = ldw(row + ja);
        if (tx < 2 * R) s_in[r * W + tx + 64] = ldw(row + jb);
    }
    __syncthreads();
    for (int r = ty; r < H; r += 4) {                            // row pass, :541-575
        const double *row = s_in + r * W + R + tx;
        double sum = taps.B[0] * row[0];
#pragma unroll
        for (int k = 1; k <= R; k++) sum += taps.B[k] * (row[-k] + row[k]);
        s_mid[r * 64 + tx] = sizeof(T) == sizeof(float) ? (double) (float) sum : sum;
    }
    __syncthreads();
    const int j = j0 + tx;
    if (j >= nx) return;
    constexpr int SEG = GXY_TH / 4;                              // rows per thread of the column pass
    double win[SEG + 2 * R];
#pragma unroll
    for (int q = 0; q < SEG + 2 * R; q++) win[q] = s_mid[(ty * SEG + q) * 64 + tx];
#pragma unroll
    for (int o = 0; o < SEG; o++) {                              // column pass, :577-611
        const int i = i0 + ty * SEG + o;
        double sum = taps.B[0] * win[o + R];
#pragma unroll
        for (int k = 1; k <= R; k++) sum += taps.B[k] * (win[o + R - k] + win[o + R + k]);
        if (i < ny) stn(out + (size_t) i * nx + j, sum);
    }
}
// zoom_out with zfactor = 1/2 (src/zoom.cpp:41-78) in ONE launch: the bicubic sample of the smoothed image at (2 j1, 2 i1) is the
// smoothed pixel itself -- at integer coordinates cubic_interpolation_cell returns v1 + 0.5 * 0 * (...) = v1 in both directions
// (src/bicubic_interpolation.cpp:108-145; the smoothed values are >= 0, so no signed zero is involved) -- and 2 j1 <= nx - 1 for
// every output column of zoom_size.  So only the even columns of the row pass and the even rows of the column pass are computed
// (same sums in the same order per pixel), written straight into the coarser level: the smoothed full-size image is neither
// stored nor re-read (26 -> 10 bytes per input pixel) and the resample launch disappears.
#define GXD_THO 12                                               // output rows per block (24 input rows + 2 R of halo)
template <typename T, int R>
__global__ __launch_bounds__(256) void k_gauss_xy_dec(OfxPlanes2<const T> in_p, OfxPlanes2<T> out_p, int nx, int ny, int nxo, int nyo,
                                                      GaussTaps taps)
{
    constexpr int W = 128 + 2 * R, H = 2 * GXD_THO + 2 * R;
    __shared__ double s_in[H * W];
    __shared__ double s_mid[H * 64];
    const T *__restrict__ in = in_p.at(blockIdx.z);
    T *__restrict__ out = out_p.at(blockIdx.z);
    const int j0 = blockIdx.x * 128, i0 = blockIdx.y * 2 * GXD_THO;       // origin of the block's input region
    const int tx = threadIdx.x, ty = threadIdx.y;
    int jc[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        int j = gauss_reflect(j0 - R + tx + 64 * q, nx);
        jc[q] = j < 0 ? 0 : (j > nx - 1 ? nx - 1 : j);
    }
    for (int r = ty; r < H; r += 4) {
        int i = gauss_reflect(i0 - R + r, ny);
        i = i < 0 ? 0 : (i > ny - 1 ? ny - 1 : i);
        const T *__restrict__ row = in + (size_t) i * nx;
        s_in[r * W + tx] = ldw(row + jc[0]);
        s_in[r * W + tx + 64] = ldw(row + jc[1]);
        if (tx < 2 * R) s_in[r * W + tx + 128] = ldw(row + jc[2]);
    }
    __syncthreads();
    for (int r = ty; r < H; r += 4) {                            // row pass at the even columns, operators.cpp:541-575
        const double *row = s_in + r * W + R + 2 * tx;
        double sum = taps.B[0] * row[0];
#pragma unroll
        for (int k = 1; k <= R; k++) sum += taps.B[k] * (row[-k] + row[k]);
        s_mid[r * 64 + tx] = sizeof(T) == sizeof(float) ? (double) (float) sum : sum;
    }
    __syncthreads();
    const int jo = blockIdx.x * 64 + tx;
    if (jo >= nxo) return;
    constexpr int SEG = GXD_THO / 4;                             // output rows per thread
#pragma unroll
    for (int o = 0; o < SEG; o++) {                              // column pass at the even rows, :577-611
        const int lo = ty * SEG + o, io = blockIdx.y * GXD_THO + lo;
        const double *c = s_mid + (R + 2 * lo) * 64 + tx;
        double sum = taps.B[0] * c[0];
#pragma unroll
        for (int k = 1; k <= R; k++) sum += taps.B[k] * (c[-k * 64] + c[k * 64]);
        if (io < nyo) stn(out + (size_t) io * nxo + jo, sum);
    }
}

template <typename T>
static void gauss_xy_launch(ofx_ctx *ctx, dim3 grid, OfxPlanes2<const T> in, OfxPlanes2<T> out, int nx, int ny, const GaussTaps &taps)
{
    const dim3 block(64, 4);
#define OFX_GXY(R_) case R_: hipLaunchKernelGGL((k_gauss_xy_gr<T, R_>), grid, block, 0, ctx->stream, in, out, nx, ny, taps); break
    switch (taps.size - 1) {
        OFX_GXY(1); OFX_GXY(2); OFX_GXY(3); OFX_GXY(4); OFX_GXY(5); OFX_GXY(6); OFX_GXY(7); OFX_GXY(8);
    default: hipLaunchKernelGGL(k_gauss_xy_g<T>, grid, block, 0, ctx->stream, in, out, nx, ny, taps); break;
    }
#undef OFX_GXY
}

template <typename T> int op_gaussian(ofx_ctx *ctx, T *I, T *tmp, int nx, int ny, double sigma, int dirichlet)
{
    GaussTaps taps;
    if (ofx_gauss_taps(sigma, &taps) != OFX_OK)
        return ofx_fail(ctx, OFX_ERR_ARG, "gaussian: sigma %g needs more than %d taps", sigma, OFX_GAUSS_MAX_TAPS);
    taps.dirichlet = dirichlet != 0;
    // reference: throws when size > xdim (:520-522); reads out of bounds when size == xdim or
    // size >= ydim -- all three are reported as OFX_ERR_SIGMA here.  (Dirichlet: no check, the padded line always has room.)
    if (!dirichlet && (taps.size >= nx || taps.size >= ny))
        return ofx_fail(ctx, OFX_ERR_SIGMA, "GaussianSmooth: sigma too large (radius %d, image %dx%d)",
                        taps.size, nx, ny);
    hipLaunchKernelGGL((k_gauss_pass<T, true>), grid2d(nx, ny), block2d(), 0, ctx->stream, (const T *) I, tmp,
                       nx, ny, taps);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((k_gauss_pass<T, false>), grid2d(nx, ny), block2d(), 0, ctx->stream, (const T *) tmp, I,
                       nx, ny, taps);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// ---- bicubic resampling: zoom_out / zoom_in (src/zoom.cpp:41-78,132-155) ---------------------------
template <typename T>
OFX_DEV void resample_px(const T *__restrict__ in, T *__restrict__ out, int nx, int ny, int nxx, int nyy, double fx, double fy)
{
    const int j1 = blockIdx.x * BX + threadIdx.x;
    const int i1 = blockIdx.y * BY + threadIdx.y;
    if (j1 >= nxx || i1 >= nyy) return;
    const double i2 = i1 / fy, j2 = j1 / fx;
    const BicubicTaps t = bicubic_taps(j2, i2, nx, ny);
    stn(out + (size_t) i1 * nxx + j1, bicubic_sample(in, t, nx));
}
template <typename T>
__global__ void k_resample(const T *__restrict__ in, T *__restrict__ out, int nx, int ny, int nxx, int nyy,
                           double fx, double fy)
{
    resample_px<T>(in, out, nx, ny, nxx, nyy, fx, fy);
}
template <typename T>
__global__ void k_resample_g(OfxPlanes2<const T> in, OfxPlanes2<T> out, int nx, int ny, int nxx, int nyy, double fx, double fy)
{
    resample_px<T>(in.at(blockIdx.z), out.at(blockIdx.z), nx, ny, nxx, nyy, fx, fy);
}

template <typename T>
__global__ void k_zoom_in_flow(const typename Pix<T>::v2 *__restrict__ U, typename Pix<T>::v2 *__restrict__ Uout,
                               int nx, int ny, int nxx, int nyy, double fx, double fy, double scale)
{
    const int j1 = blockIdx.x * BX + threadIdx.x;
    const int i1 = blockIdx.y * BY + threadIdx.y;
    if (j1 >= nxx || i1 >= nyy) return;
    const double i2 = i1 / fy, j2 = j1 / fx;
    const BicubicTaps t = bicubic_taps(j2, i2, nx, ny);
    double c1[4], c2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double2 v0 = ldw2(U + (size_t) t.row[0] * nx + t.col[k]);
        const double2 v1 = ldw2(U + (size_t) t.row[1] * nx + t.col[k]);
        const double2 v2 = ldw2(U + (size_t) t.row[2] * nx + t.col[k]);
        const double2 v3 = ldw2(U + (size_t) t.row[3] * nx + t.col[k]);
        c1[k] = cubic_cell(v0.x, v1.x, v2.x, v3.x, t.fy);
        c2[k] = cubic_cell(v0.y, v1.y, v2.y, v3.y, t.fy);
    }
    double2 r;
    // In float storage the reference-equivalent value would be rounded to T between zoom_in and the
    // `*= 1/zfactor` loop; in double storage both are exact IEEE steps, so one fused store is identical.
    r.x = cubic_cell(c1[0], c1[1], c1[2], c1[3], t.fx) * scale;
    r.y = cubic_cell(c2[0], c2[1], c2[2], c2[3], t.fx) * scale;
    stn2(Uout + (size_t) i1 * nxx + j1, r);
}

template <typename T>
int op_resample(ofx_ctx *ctx, const T *in, T *out, int nx, int ny, int nxx, int nyy, double fx, double fy)
{
    hipLaunchKernelGGL(k_resample<T>, grid2d(nxx, nyy), block2d(), 0, ctx->stream, in, out, nx, ny, nxx, nyy, fx, fy);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

template <typename T>
int op_zoom_in_flow(ofx_ctx *ctx, const typename Pix<T>::v2 *U, typename Pix<T>::v2 *Uout, int nx, int ny,
                    int nxx, int nyy, double scale)
{
    const double fx = ((double) nxx / nx), fy = ((double) nyy / ny);     // zoom.cpp:141-142
    hipLaunchKernelGGL(k_zoom_in_flow<T>, grid2d(nxx, nyy), block2d(), 0, ctx->stream, U, Uout, nx, ny, nxx, nyy,
                       fx, fy, scale);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

template <typename T>
int op_zoom_out(ofx_ctx *ctx, const T *I, T *Iout, T *tmpA, T *tmpB, int nx, int ny, double factor)
{
    int nxx, nyy;
    ofx_zoom_size(nx, ny, &nxx, &nyy, factor);
    const double sigma = 0.6 * sqrt(1.0 / (factor * factor) - 1.0);      // ZOOM_SIGMA_ZERO, zoom.cpp:15,60
    OFX_HIP(ctx, hipMemcpyAsync(tmpA, I, (size_t) nx * ny * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    OFX_TRY(op_gaussian<T>(ctx, tmpA, tmpB, nx, ny, sigma));
    return op_resample<T>(ctx, tmpA, Iout, nx, ny, nxx, nyy, factor, factor);
}

// ---- planar stencil operators (operator-level API) --------------------------------------------------
template <typename T>
__global__ void k_divergence(const T *__restrict__ v1, const T *__restrict__ v2, T *__restrict__ div, int nx, int ny)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    const double ac = ldw(v1 + p), bc = ldw(v2 + p);
    const double al = j > 0 ? ldw(v1 + p - 1) : 0.0;
    const double bu = i > 0 ? ldw(v2 + p - nx) : 0.0;
    stn(div + p, div_backward(ac, al, bc, bu, j == 0, j == nx - 1, i == 0, i == ny - 1));
}

// src/operators.cpp:86-125
template <typename T>
__global__ void k_forward_gradient(const T *__restrict__ f, T *__restrict__ fx, T *__restrict__ fy, int nx, int ny)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    const double c = ldw(f + p);
    stn(fx + p, j < nx - 1 ? ldw(f + p + 1) - c : 0.0);
    stn(fy + p, i < ny - 1 ? ldw(f + p + nx) - c : 0.0);
}

// src/operators.cpp:335-406 (nz = 1): the missing neighbour at a border is the pixel itself, factor 1/2 kept
template <typename T>
__global__ void k_centered_gradient(const T *__restrict__ f, T *__restrict__ dx, T *__restrict__ dy, int nx, int ny)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const size_t p = (size_t) i * nx + j;
    stn(dx + p, 0.5 * (ldw(f + (size_t) i * nx + jr) - ldw(f + (size_t) i * nx + jl)));
    stn(dy + p, 0.5 * (ldw(f + (size_t) id * nx + j) - ldw(f + (size_t) iu * nx + j)));
}

// Dxx / Dyy / Dxy = mask3x3 (src/operators.cpp:132-328) specialised to the three fixed masks.  Taps
// that fall outside fold onto the edge sample and their weights are summed before the multiply, so
// e.g. Dxx at j=0 is in[0]*(1-2) + in[1]; zero-weight taps add +-0 and drop out.
template <typename T>
__global__ void k_second_derivative(const T *__restrict__ f, T *__restrict__ out, int nx, int ny, int which)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    double r;
    if (which == 0) {                       // Dxx: 1 -2 1 along x
        if (j == 0)            r = ldw(f + p) * -1.0 + ldw(f + p + 1);
        else if (j == nx - 1)  r = ldw(f + p - 1) + ldw(f + p) * -1.0;
        else                   r = ldw(f + p - 1) + ldw(f + p) * -2.0 + ldw(f + p + 1);
    } else if (which == 1) {                // Dyy: 1 -2 1 along y
        if (i == 0)            r = ldw(f + p) * -1.0 + ldw(f + p + nx);
        else if (i == ny - 1)  r = ldw(f + p - nx) + ldw(f + p) * -1.0;
        else                   r = ldw(f + p - nx) + ldw(f + p) * -2.0 + ldw(f + p + nx);
    } else {                                // Dxy: +-1/4 on the four diagonal neighbours (clamped)
        const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
        const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
        r = ldw(f + (size_t) iu * nx + jl) * 0.25 + ldw(f + (size_t) iu * nx + jr) * -0.25 +
            ldw(f + (size_t) id * nx + jl) * -0.25 + ldw(f + (size_t) id * nx + jr) * 0.25;
    }
    stn(out + p, r);
}

// src/bicubic_interpolation.cpp:352-374
template <typename T>
__global__ void k_bicubic_warp(const T *__restrict__ in, const T *__restrict__ u, const T *__restrict__ v,
                               T *__restrict__ out, int nx, int ny, int border_out)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t p = (size_t) i * nx + j;
    const double uu = j + ldw(u + p), vv = i + ldw(v + p);
    const BicubicTaps t = bicubic_taps(uu, vv, nx, ny);
    stn(out + p, (t.out && border_out) ? 0.0 : bicubic_sample(in, t, nx));
}

template <typename T>
__global__ void k_bicubic_at(const T *__restrict__ in, const double *__restrict__ uu, const double *__restrict__ vv,
                             double *__restrict__ out, int n, int nx, int ny, int border_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const BicubicTaps t = bicubic_taps(uu[k], vv[k], nx, ny);
    out[k] = (t.out && border_out) ? 0.0 : bicubic_sample(in, t, nx);
}

// bicubic_interpolation_at_color (src/bicubic_interpolation.cpp:253-344): channel k of an image with nz interleaved
// channels, same tap rule as the scalar version
template <typename T>
__global__ void k_bicubic_at_color(const T *__restrict__ in, const double *__restrict__ uu, const double *__restrict__ vv,
                                   double *__restrict__ out, int n, int nx, int ny, int nz, int ch, int border_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const BicubicTaps t = bicubic_taps(uu[k], vv[k], nx, ny);
    double r = 0.0;
    if (!(t.out && border_out)) {
        double c[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double v0 = ldw(in + ((size_t) t.row[0] * nx + t.col[q]) * nz + ch);
            const double v1 = ldw(in + ((size_t) t.row[1] * nx + t.col[q]) * nz + ch);
            const double v2 = ldw(in + ((size_t) t.row[2] * nx + t.col[q]) * nz + ch);
            const double v3 = ldw(in + ((size_t) t.row[3] * nx + t.col[q]) * nz + ch);
            c[q] = cubic_cell(v0, v1, v2, v3, t.fy);
        }
        r = cubic_cell(c[0], c[1], c[2], c[3], t.fx);
    }
    out[k] = r;
}

// the temporal part of centered_gradient3 (src/operators.cpp:477-498): centred difference between frames,
// one-sided (still x 0.5) in the first / last frame, 0 for a single frame
template <typename T>
__global__ void k_gradient_dz(const T *__restrict__ in, T *__restrict__ dz, size_t df, int nz)
{
    const size_t p = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= df) return;
    const size_t k = (size_t) f * df + p;
    double r = 0.0;
    if (nz > 1) {
        const double hi = ldw(in + (f < nz - 1 ? k + df : k)), lo = ldw(in + (f > 0 ? k - df : k));
        r = 0.5 * (hi - lo);
    }
    stn(dz + k, r);
}

template <typename T> int op_divergence(ofx_ctx *ctx, const T *v1, const T *v2, T *div, int nx, int ny)
{
    hipLaunchKernelGGL(k_divergence<T>, grid2d(nx, ny), block2d(), 0, ctx->stream, v1, v2, div, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_forward_gradient(ofx_ctx *ctx, const T *f, T *fx, T *fy, int nx, int ny)
{
    hipLaunchKernelGGL(k_forward_gradient<T>, grid2d(nx, ny), block2d(), 0, ctx->stream, f, fx, fy, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_centered_gradient(ofx_ctx *ctx, const T *f, T *dx, T *dy, int nx, int ny)
{
    hipLaunchKernelGGL(k_centered_gradient<T>, grid2d(nx, ny), block2d(), 0, ctx->stream, f, dx, dy, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T> int op_second_derivative(ofx_ctx *ctx, const T *f, T *out, int nx, int ny, int which)
{
    hipLaunchKernelGGL(k_second_derivative<T>, grid2d(nx, ny), block2d(), 0, ctx->stream, f, out, nx, ny, which);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T>
int op_bicubic_warp(ofx_ctx *ctx, const T *in, const T *u, const T *v, T *out, int nx, int ny, int border_out)
{
    hipLaunchKernelGGL(k_bicubic_warp<T>, grid2d(nx, ny), block2d(), 0, ctx->stream, in, u, v, out, nx, ny, border_out);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}
template <typename T>
int op_bicubic_at(ofx_ctx *ctx, const T *in, const double *uu, const double *vv, double *out, int n, int nx,
                  int ny, int border_out)
{
    hipLaunchKernelGGL(k_bicubic_at<T>, dim3(grid1d((size_t) n)), dim3(256), 0, ctx->stream, in, uu, vv, out, n, nx,
                       ny, border_out);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// centred gradient next to the image (src/operators.cpp:335-406, nz = 1): pa = (f, fx), pb = fy
template <typename T>
__global__ void k_grad_pack(const T *__restrict__ f, typename Pix<T>::v2 *__restrict__ pa, T *__restrict__ pb, int nx,
                            int ny)
{
    const int j = blockIdx.x * BX + threadIdx.x;
    const int i = blockIdx.y * BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t zoff = (size_t) blockIdx.z * nx * ny;      // plane of a lockstep group
    f += zoff;
    pa += zoff;
    pb += zoff;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const size_t p = (size_t) i * nx + j;
    const double fx = 0.5 * (ldw(f + (size_t) i * nx + jr) - ldw(f + (size_t) i * nx + jl));
    const double fy = 0.5 * (ldw(f + (size_t) id * nx + j) - ldw(f + (size_t) iu * nx + j));
    stn2(pa + p, make_double2(ldw(f + p), fx));
    stn(pb + p, fy);
}

// G planes back to back (blockIdx.z = plane)
template <typename T> int op_grad_pack(ofx_ctx *ctx, const T *f, typename Pix<T>::v2 *pa, T *pb, int nx, int ny, int G)
{
    dim3 g = grid2d(nx, ny);
    g.z = G;
    hipLaunchKernelGGL(k_grad_pack<T>, g, block2d(), 0, ctx->stream, f, pa, pb, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

int op_pyramid_sizes(ofx_ctx *ctx, int nxx, int nyy, int nscales, double zfactor, std::vector<int> &nxs, std::vector<int> &nys)
{
    // no fixed limit on the number of levels (the reference allocates its pyramid dynamically; tvl1flow with
    // zfactor = 0.9 at 1080p uses 47): OFX_MAX_SCALES only bounds what ofx_stats RECORDS.  1024 is a sanity bound.
    if (nscales < 1 || nscales > 1024) return ofx_fail(ctx, OFX_ERR_ARG, "nscales=%d", nscales);
    nxs.assign(nscales, 0);
    nys.assign(nscales, 0);
    if (nxx < 2 || nyy < 2) return ofx_fail(ctx, OFX_ERR_ARG, "image %dx%d too small", nxx, nyy);
    if (!(zfactor > 0.0) || !(zfactor < 1.0)) return ofx_fail(ctx, OFX_ERR_ARG, "zoom factor %g", zfactor);
    int nx = nxx, ny = nyy;
    for (int s = 0; s < nscales; s++) {
        if (s) ofx_zoom_size(nxs[s - 1], nys[s - 1], &nx, &ny, zfactor);
        if (nx < 2 || ny < 2) return ofx_fail(ctx, OFX_ERR_SIGMA, "scale %d would be %dx%d", s, nx, ny);
        nxs[s] = nx;
        nys[s] = ny;
    }
    return OFX_OK;
}

size_t op_pyramid_scratch_doubles() { return (size_t) MM_SCR; }

// lv[s] = sizes and DESTINATION arrays of every level (caller-allocated); tmpA / tmpB: full-size scratch
// images, scr: op_pyramid_scratch_doubles() doubles.
template <typename T>
int op_build_pyramid_into(ofx_ctx *ctx, const T *dA, const T *dB, int nscales, double zfactor, double sigma,
                          const std::vector<ImgLevel<T>> &lv, T *tmpA, T *tmpB, double *scr)
{
    const int nxx = lv[0].nx, nyy = lv[0].ny;
    OFX_TRY(op_normalize2<T>(ctx, dA, dB, lv[0].A, lv[0].B, nxx * nyy, scr));
    OFX_TRY(op_gaussian<T>(ctx, lv[0].A, tmpA, nxx, nyy, sigma));
    OFX_TRY(op_gaussian<T>(ctx, lv[0].B, tmpA, nxx, nyy, sigma));
    for (int s = 1; s < nscales; s++) {
        OFX_TRY(op_zoom_out<T>(ctx, lv[s - 1].A, lv[s].A, tmpA, tmpB, lv[s - 1].nx, lv[s - 1].ny, zfactor));
        OFX_TRY(op_zoom_out<T>(ctx, lv[s - 1].B, lv[s].B, tmpA, tmpB, lv[s - 1].nx, lv[s - 1].ny, zfactor));
    }
    return OFX_OK;
}

// The same prologue for the G pairs of a lockstep group with ONE launch per step for all 2 G images (per pair it is 39
// launches, most of them tiny): lvA[s] / lvB[s] = level arrays holding the G images back to back, tmpA / tmpB = scratch
// for 2 G full-size images each, scr = G * op_pyramid_scratch_doubles() doubles.  Same per-pixel arithmetic as
// op_build_pyramid_into; zoom_out's scratch copy (zoom.cpp:54-57) is not made -- the first Gaussian pass reads the level itself.
template <typename T>
int op_build_pyramid_group(ofx_ctx *ctx, int G, const void *const *dA, const void *const *dB, int nscales, double zfactor,
                           double sigma, const int *nxs, const int *nys, T *const *lvA, T *const *lvB, T *tmpA, T *tmpB,
                           double *scr)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "pyramid group of %d", G);
    const int nxx = nxs[0], nyy = nys[0], size = nxx * nyy;
    OfxGroupPtrs P;
    for (int g = 0; g < OFX_MAX_GROUP; g++) { P.a[g] = g < G ? dA[g] : nullptr; P.b[g] = g < G ? dB[g] : nullptr; }
    int nb = grid1d((size_t) size);
    if (nb > MM_BLOCKS) nb = MM_BLOCKS;
    hipLaunchKernelGGL(k_minmax_partial_g<T>, dim3(nb, G), dim3(256), 0, ctx->stream, P, size, scr);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_minmax_final_g, dim3(G), dim3(256), 0, ctx->stream, scr, nb);
    OFX_LAUNCH_CHECK(ctx);
    // fused row + column pass (k_gauss_xy_g) when the radius allows: it cannot work in place, so the normalised images go to
    // the scratch array and the presmoothing writes the level
    GaussTaps t0;
    const double zsigma = 0.6 * sqrt(1.0 / (zfactor * zfactor) - 1.0);          // ZOOM_SIGMA_ZERO, zoom.cpp:15,60
    bool fused = ctx->gauss_fused != 0 && ofx_gauss_taps(sigma, &t0) == OFX_OK && t0.size - 1 <= GXY_RMAX;
    if (fused && nscales > 1) fused = ofx_gauss_taps(zsigma, &t0) == OFX_OK && t0.size - 1 <= GXY_RMAX;
    T *n0 = fused ? tmpA : lvA[0], *n1 = fused ? tmpA + (size_t) G * size : lvB[0];
    hipLaunchKernelGGL(k_normalize2_g<T>, dim3(grid1d((size_t) size), G), dim3(256), 0, ctx->stream, P, n0, n1, size,
                       (const double *) scr);
    OFX_LAUNCH_CHECK(ctx);
    auto gauss_xy = [&](const T *ia, const T *ib, T *oa, T *ob, int nx, int ny, double sg) -> int {
        GaussTaps taps;
        if (ofx_gauss_taps(sg, &taps) != OFX_OK)
            return ofx_fail(ctx, OFX_ERR_ARG, "gaussian: sigma %g needs more than %d taps", sg, OFX_GAUSS_MAX_TAPS);
        if (taps.size >= nx || taps.size >= ny)
            return ofx_fail(ctx, OFX_ERR_SIGMA, "GaussianSmooth: sigma too large (radius %d, image %dx%d)", taps.size, nx, ny);
        const size_t st = (size_t) nx * ny;
        if (ctx->gauss_fused == 2)                         // the generic-radius kernel (A/B and tests)
            hipLaunchKernelGGL(k_gauss_xy_g<T>, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, GXY_TH), 2 * G), dim3(64, 4), 0, ctx->stream,
                               OfxPlanes2<const T>{ia, ib, G, st}, OfxPlanes2<T>{oa, ob, G, st}, nx, ny, taps);
        else
            gauss_xy_launch<T>(ctx, dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, GXY_TH), 2 * G), OfxPlanes2<const T>{ia, ib, G, st},
                               OfxPlanes2<T>{oa, ob, G, st}, nx, ny, taps);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    if (fused) {
        OFX_TRY(gauss_xy(n0, n1, lvA[0], lvB[0], nxx, nyy, sigma));
        for (int s = 1; s < nscales; s++) {
            const int nx = nxs[s - 1], ny = nys[s - 1];
            const size_t st = (size_t) nx * ny;
            GaussTaps tz;
            if (ctx->gauss_fused == 1 && zfactor == 0.5 && ofx_gauss_taps(zsigma, &tz) == OFX_OK && tz.size - 1 == 5 && tz.size < nx &&
                tz.size < ny && 2 * (nxs[s] - 1) <= nx - 1 && 2 * (nys[s] - 1) <= ny - 1) {
                // smoothing and the 2:1 sampling in one launch (k_gauss_xy_dec)
                hipLaunchKernelGGL((k_gauss_xy_dec<T, 5>), dim3(ofx_cdiv(nxs[s], 64), ofx_cdiv(nys[s], GXD_THO), 2 * G), dim3(64, 4), 0,
                                   ctx->stream, OfxPlanes2<const T>{lvA[s - 1], lvB[s - 1], G, st},
                                   OfxPlanes2<T>{lvA[s], lvB[s], G, (size_t) nxs[s] * nys[s]}, nx, ny, nxs[s], nys[s], tz);
                OFX_LAUNCH_CHECK(ctx);
                continue;
            }
            OFX_TRY(gauss_xy(lvA[s - 1], lvB[s - 1], tmpB, tmpB + (size_t) G * st, nx, ny, zsigma));
            dim3 g = grid2d(nxs[s], nys[s]);
            g.z = 2 * G;
            hipLaunchKernelGGL(k_resample_g<T>, g, block2d(), 0, ctx->stream, OfxPlanes2<const T>{tmpB, tmpB + (size_t) G * st, G, st},
                               OfxPlanes2<T>{lvA[s], lvB[s], G, (size_t) nxs[s] * nys[s]}, nx, ny, nxs[s], nys[s], zfactor, zfactor);
            OFX_LAUNCH_CHECK(ctx);
        }
        return OFX_OK;
    }
    auto gauss = [&](const T *ia, const T *ib, T *oa, T *ob, T *ta, T *tb, int nx, int ny, double sg) -> int {
        GaussTaps taps;
        if (ofx_gauss_taps(sg, &taps) != OFX_OK)
            return ofx_fail(ctx, OFX_ERR_ARG, "gaussian: sigma %g needs more than %d taps", sg, OFX_GAUSS_MAX_TAPS);
        if (taps.size >= nx || taps.size >= ny)
            return ofx_fail(ctx, OFX_ERR_SIGMA, "GaussianSmooth: sigma too large (radius %d, image %dx%d)", taps.size, nx, ny);
        const size_t st = (size_t) nx * ny;
        dim3 g = grid2d(nx, ny);
        g.z = 2 * G;
        hipLaunchKernelGGL((k_gauss_pass_g<T, true>), g, block2d(), 0, ctx->stream, OfxPlanes2<const T>{ia, ib, G, st},
                           OfxPlanes2<T>{ta, tb, G, st}, nx, ny, taps);
        OFX_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((k_gauss_pass_g<T, false>), g, block2d(), 0, ctx->stream, OfxPlanes2<const T>{ta, tb, G, st},
                           OfxPlanes2<T>{oa, ob, G, st}, nx, ny, taps);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    // scratch: first half = the A images of the group, second half = the B images
    OFX_TRY(gauss(lvA[0], lvB[0], lvA[0], lvB[0], tmpA, tmpA + (size_t) G * size, nxx, nyy, sigma));
    for (int s = 1; s < nscales; s++) {
        const int nx = nxs[s - 1], ny = nys[s - 1];
        const size_t st = (size_t) nx * ny;
        OFX_TRY(gauss(lvA[s - 1], lvB[s - 1], tmpB, tmpB + (size_t) G * st, tmpA, tmpA + (size_t) G * st, nx, ny, zsigma));
        dim3 g = grid2d(nxs[s], nys[s]);
        g.z = 2 * G;
        hipLaunchKernelGGL(k_resample_g<T>, g, block2d(), 0, ctx->stream,
                           OfxPlanes2<const T>{tmpB, tmpB + (size_t) G * st, G, st},
                           OfxPlanes2<T>{lvA[s], lvB[s], G, (size_t) nxs[s] * nys[s]}, nx, ny, nxs[s], nys[s], zfactor, zfactor);
        OFX_LAUNCH_CHECK(ctx);
    }
    return OFX_OK;
}

template <typename T>
int op_build_pyramid(ofx_ctx *ctx, const T *dA, const T *dB, int nxx, int nyy, int nscales, double zfactor,
                     double sigma, std::vector<ImgLevel<T>> &lv)
{
    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nxx, nyy, nscales, zfactor, nxs, nys));
    lv.resize(nscales);
    for (int s = 0; s < nscales; s++) {
        lv[s].nx = nxs[s];
        lv[s].ny = nys[s];
        OFX_TRY(ofx_alloc(ctx, (size_t) nxs[s] * nys[s], &lv[s].A));
        OFX_TRY(ofx_alloc(ctx, (size_t) nxs[s] * nys[s], &lv[s].B));
    }
    T *tmpA, *tmpB;
    double *scr;
    OFX_TRY(ofx_alloc(ctx, (size_t) nxx * nyy, &tmpA));
    OFX_TRY(ofx_alloc(ctx, (size_t) nxx * nyy, &tmpB));
    OFX_TRY(ofx_alloc(ctx, op_pyramid_scratch_doubles(), &scr));
    return op_build_pyramid_into<T>(ctx, dA, dB, nscales, zfactor, sigma, lv, tmpA, tmpB, scr);
}

template <typename T>
int op_bicubic_at_color(ofx_ctx *ctx, const T *in, const double *uu, const double *vv, double *out, int n, int nx, int ny,
                        int nz, int ch, int border_out)
{
    hipLaunchKernelGGL(k_bicubic_at_color<T>, dim3(grid1d((size_t) n)), dim3(256), 0, ctx->stream, in, uu, vv, out, n, nx,
                       ny, nz, ch, border_out);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

template <typename T> int op_gradient_dz(ofx_ctx *ctx, const T *in, T *dz, int nx, int ny, int nz)
{
    const size_t df = (size_t) nx * ny;
    hipLaunchKernelGGL(k_gradient_dz<T>, dim3(grid1d(df), nz), dim3(256), 0, ctx->stream, in, dz, df, nz);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// getminmax (src/utils.cpp:509-525) of one array: mm[0] = min, mm[1] = max on the device
template <typename T> int op_minmax(ofx_ctx *ctx, const T *x, int size, double *scr)
{
    int nb = grid1d((size_t) size);
    if (nb > MM_BLOCKS) nb = MM_BLOCKS;
    hipLaunchKernelGGL(k_minmax_partial<T>, dim3(nb), dim3(256), 0, ctx->stream, x, x, size, scr);
    OFX_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, ctx->stream, scr, nb, scr + 2 * MM_BLOCKS);
    OFX_LAUNCH_CHECK(ctx);
    return OFX_OK;
}

// ---- explicit instantiations -----------------------------------------------------------------------
#define OFX_INSTANTIATE(T)                                                                                           \
    template int op_convert_in<T>(ofx_ctx *, const double *, T *, size_t);                                            \
    template int op_convert_out<T>(ofx_ctx *, const T *, double *, size_t);                                           \
    template int op_interleave2<T>(ofx_ctx *, const double *, const double *, Pix<T>::v2 *, size_t);                  \
    template int op_deinterleave2<T>(ofx_ctx *, const Pix<T>::v2 *, double *, double *, size_t);                      \
    template int op_to_flo<T>(ofx_ctx *, const Pix<T>::v2 *, float2 *, size_t);                                       \
    template int op_fill2<T>(ofx_ctx *, Pix<T>::v2 *, size_t);                                                        \
    template int op_normalize2<T>(ofx_ctx *, const T *, const T *, T *, T *, int, double *);                          \
    template int op_gaussian<T>(ofx_ctx *, T *, T *, int, int, double, int);                                          \
    template int op_resample<T>(ofx_ctx *, const T *, T *, int, int, int, int, double, double);                       \
    template int op_zoom_in_flow<T>(ofx_ctx *, const Pix<T>::v2 *, Pix<T>::v2 *, int, int, int, int, double);         \
    template int op_zoom_out<T>(ofx_ctx *, const T *, T *, T *, T *, int, int, double);                               \
    template int op_divergence<T>(ofx_ctx *, const T *, const T *, T *, int, int);                                    \
    template int op_forward_gradient<T>(ofx_ctx *, const T *, T *, T *, int, int);                                    \
    template int op_centered_gradient<T>(ofx_ctx *, const T *, T *, T *, int, int);                                   \
    template int op_second_derivative<T>(ofx_ctx *, const T *, T *, int, int, int);                                   \
    template int op_bicubic_warp<T>(ofx_ctx *, const T *, const T *, const T *, T *, int, int, int);                  \
    template int op_bicubic_at<T>(ofx_ctx *, const T *, const double *, const double *, double *, int, int, int, int);           \
    template int op_grad_pack<T>(ofx_ctx *, const T *, Pix<T>::v2 *, T *, int, int, int);                                       \
    template int op_build_pyramid_group<T>(ofx_ctx *, int, const void *const *, const void *const *, int, double, double,       \
                                           const int *, const int *, T *const *, T *const *, T *, T *, double *);              \
    template int op_bicubic_at_color<T>(ofx_ctx *, const T *, const double *, const double *, double *, int, int, int, int, int, int); \
    template int op_gradient_dz<T>(ofx_ctx *, const T *, T *, int, int, int);                                                   \
    template int op_minmax<T>(ofx_ctx *, const T *, int, double *);                                                             \
    template int op_build_pyramid<T>(ofx_ctx *, const T *, const T *, int, int, int, double, double,                   \
                                     std::vector<ImgLevel<T>> &);                                                      \
    template int op_build_pyramid_into<T>(ofx_ctx *, const T *, const T *, int, double, double,                        \
                                          const std::vector<ImgLevel<T>> &, T *, T *, double *);

OFX_INSTANTIATE(double)
OFX_INSTANTIATE(float)
