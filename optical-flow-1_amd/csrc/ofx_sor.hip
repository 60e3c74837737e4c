// ofx_sor.hip -- Horn-Schunck pyramidal and Brox spatial on gfx950
// (reference: src/horn_schunck_pyramidal.cpp, src/brox_optic_flow_spatial.cpp, src/brox_spatial_mask.cpp).
//
// Both reference solvers are in-place SOR (w = 1.9) swept in lexicographic order -- a sequential
// recurrence.  The GPU sweeps in colours instead: Horn-Schunck's 8-neighbour stencil needs four
// colours (i%2, j%2), Brox's 5-point stencil two (red-black).  Within one colour no pixel reads
// another pixel of the same colour (borders use clamped neighbour indices, which is what the
// reference's replicated border indices amount to), so the in-place update is race-free and
// deterministic.  Per-pixel arithmetic is the reference's expression for expression; what changes
// is the ORDER in which pixels see each other's new values, hence the trajectory and the sweep at
// which `error > TOL` fails.  Measured against the lexicographic reference (tests):
// Horn-Schunck AEPE 1e-5..5e-5 (inside the 1e-4 bar), Brox up to 1.6e-4 (the reference's own
// 1-vs-8-thread spread is 3.6e-5).  oracle's `set_sor_order(1)` reproduces this file bit for bit.
//
// Storage per level (T = storage type):
//   HS  : U=(u,v) pairs in place; pa=(I2,I2x), pb=I2y; A=(I2wx,I2wy); Dif=dif  -- the five coefficient
//         arrays Au,Av,Du,Dv,D of the reference are recomputed from A, Dif in registers (same ops).
//   Brox: U; G1=(I1x,I1y); PA=(I2,I2x,I2y,I2xx), PB=(I2xy,I2yy); WA/WB = their warps; Psis;
//         DV=(div_u,div_v), Dd=div_d; CO=(Au,Av,Du,Dv), Dm=D; DU=(du,dv) in place.
#include "ofx_ops.h"
#include "ofx_device.h"
#include "ofx_loop.h"
#include "ofx_sor_tile.h"

#include <algorithm>

#include <atomic>
#include <cmath>
#include <thread>

#define HS_SOR_W 1.9                 // src/horn_schunck_pyramidal.cpp:21
#define HS_PLANE_C_SKEW 2             // Horn-Schunck hyperplanes: pos = 2 i + j (k_hs_plane)
#define BROX_PLANE_C_SKEW 1           // Brox hyperplanes: pos = i + j (k_brox_plane)
#define HS_PRESMOOTH_SIGMA 0.8       // src/horn_schunck_pyramidal.cpp:22
#define BROX_EPSILON 0.001           // src/brox_optic_flow_spatial.cpp:23
#define BROX_SOR_W 1.9               // src/brox_optic_flow_spatial.cpp:25
#define BROX_SIGMA 0.8               // src/brox_optic_flow_spatial.cpp:26

template <typename T> OFX_DEV double rnd_to(double x);
template <> OFX_DEV double rnd_to<double>(double x) { return x; }
template <> OFX_DEV double rnd_to<float>(double x) { return (double) (float) x; }

static inline dim3 g2d(int nx, int ny) { return dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4)); }
static inline dim3 b2d() { return dim3(64, 4); }

static int sor_pick_chunk(const ofx_ctx *ctx, int nx, int ny, int launches_per_sweep)
{
    if (ctx->chunk > 0) return ctx->chunk;
    const double est_us = fmax(4.0 * launches_per_sweep, (double) nx * ny * 100.0 / 2.0e6);
    int c = (int) (60.0 / est_us);
    return c < 2 ? 2 : (c > 32 ? 32 : c);
}

// ============================================================================================
// Horn-Schunck
// ============================================================================================

// warp of (I2, I2x, I2y) + constant parts of the system, src/horn_schunck_pyramidal.cpp:123-137
// (ucode: bit g set = the flow of pair g lives in U1, the second buffer of the tile sweeps' ping-pong, ofx_sor_tile.hip)
template <typename T>
__global__ void k_hs_warp(const typename Pix<T>::v2 *__restrict__ pa, const T *__restrict__ pb, const T *__restrict__ I1,
                          const typename Pix<T>::v2 *__restrict__ U, const typename Pix<T>::v2 *__restrict__ U1, unsigned ucode,
                          typename Pix<T>::v2 *__restrict__ A, T *__restrict__ Dif, int nx, int ny)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t go = (size_t) blockIdx.z * nx * ny;             // pair of a lockstep group
    if ((ucode >> blockIdx.z) & 1u) U = U1;
    pa += go; pb += go; I1 += go; U += go; A += go; Dif += go;
    const size_t p = (size_t) i * nx + j;
    const double2 u = ldw2(U + p);
    const BicubicTaps t = bicubic_taps(j + u.x, i + u.y, nx, ny);
    double I2w = 0.0, I2wx = 0.0, I2wy = 0.0;
    if (!t.out) {
        bicubic_sample3(pa, pb, t, nx, I2w, I2wx, I2wy);
        I2wx = rnd_to<T>(I2wx);
        I2wy = rnd_to<T>(I2wy);
    }
    const double I2wl = I2wx * u.x + I2wy * u.y;                 // :130
    const double dif = ldw(I1 + p) - I2w + I2wl;                 // :131
    stn2(A + p, make_double2(I2wx, I2wy));
    stn(Dif + p, dif);
}

// ---- how a sweep kernel reaches the in-place unknowns -------------------------------------------------------
// UGlobal: straight from / to the global array (COH: past the L1, see ldu2; SNAP: also into the sweep's snapshot).
// (An accessor that served same-launch hand-offs from an LDS ring was built and measured: correct, but slower --
// a windowed step is bound by the memory latency of the PREVIOUS sweep's values and by one CU's f64 rate, not by
// the store drain the ring removes.  DESIGN.md 5.3.)
// ---- where pixel (i, j) of a plane lives --------------------------------------------------------------------------
// LayRow: row-major, the layout of every array outside the windowed sweeps.
// LaySkew: hyperplane-major.  In the windowed exact sweeps the lanes of a wave are consecutive ROWS of one hyperplane
// (pixel (r, q - c r) for lane r), so in a row-major array every lane of every load sits in a different cache line: 64
// lines per wave instruction, 13 memory instructions per update -- the vector L1 (one line per clock and CU) was the
// bound, and a launch took 16x as long for 16 lockstep pairs (profiles/r02_c_*).  With element (i, j) at
// (c i + j) ny + i the pixels of a hyperplane are contiguous in i, and so are all their neighbours (each lies on a
// hyperplane q + const at row i + const): every access of a wave is one contiguous run.  The plane is padded to
// (c (ny - 1) + nx) * ny elements; c = 2 for the 8-neighbour stencil of Horn-Schunck, 1 for Brox.
struct LayRow {
    int nx;
    OFX_DEV size_t idx(int i, int j) const { return (size_t) i * nx + j; }
};
struct LaySkew {
    int ny, c;
    OFX_DEV size_t idx(int i, int j) const { return (size_t) (c * i + j) * ny + i; }
};
static inline size_t skew_plane_elems(int nx, int ny, int c) { return (size_t) (c * (ny - 1) + nx) * ny; }

template <typename T, bool COH, bool SNAP, class Lay = LayRow> struct UGlobal {
    typename Pix<T>::v2 *U, *snap;
    Lay lay;
    OFX_DEV double2 get(int ii, int jj) const { return ldu2<COH>(U + lay.idx(ii, jj)); }
    // the constant operands of pixel (i, j): HS (A, dif)
    OFX_DEV void coef(const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif, int i, int j, double2 &a, double &dif) const
    {
        const size_t p = lay.idx(i, j);
        a = ldw2(A + p);
        dif = ldw(Dif + p);
    }
    // Brox: psi_s of pixel (ii, jj); (CO, Dm) of pixel (i, j)
    OFX_DEV double psi(const T *__restrict__ Psis, int ii, int jj) const { return ldw(Psis + lay.idx(ii, jj)); }
    OFX_DEV void coef4(const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm, int i, int j, double4 &co, double &D) const
    {
        const size_t p = lay.idx(i, j);
        co = ldw4(CO + p);
        D = ldw(Dm + p);
    }
    OFX_DEV void put(int i, int j, double2 v) const
    {
        const size_t p = lay.idx(i, j);
        stn2(U + p, v);
        if (SNAP) stn2(snap + p, v);
    }
};

// row-major <-> skewed copies of one array of a lockstep group (blockIdx.z = pair): IN = true writes dst[skew] =
// src[row-major], false the other way.  Once per solve and array, next to tens of sweeps.
template <typename V, bool IN>
__global__ void k_skew(const V *__restrict__ src, V *__restrict__ dst, int nx, int ny, int c, size_t plane_s)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const LaySkew lay = {ny, c};
    const size_t rm = (size_t) blockIdx.z * nx * ny + (size_t) i * nx + j, sk = blockIdx.z * plane_s + lay.idx(i, j);
    if (IN) dst[sk] = src[rm];
    else dst[rm] = src[sk];
}
template <typename V, bool IN>
static int op_skew(ofx_ctx *ctx, const V *src, V *dst, int nx, int ny, int c, int G)
{
    hipLaunchKernelGGL((k_skew<V, IN>), dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G), dim3(64, 4), 0, ctx->stream, src, dst, nx,
                       ny, c, skew_plane_elems(nx, ny, c));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "skew copy launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}
// SOR update of one pixel, src/horn_schunck_pyramidal.cpp:31-71.  Neighbour indices are the clamped
// coordinates -- exactly what the reference's replicated border indices are (:161-228) -- in the order
// up-left, up-right, bottom-left, bottom-right / up, left, bottom, right, with ONE quirk kept for
// bit-exactness: the bottom-right corner lists its diagonal taps bottom pair first (:222-228).
// Returns the squared update (:70).
// operands of one update, loaded first so that several independent updates can have their loads in flight together
struct HsOps {
    double2 p1, p2, p3, p4, p5, p6, p7, p8, c, a;
    double  dif;
};
template <typename T, class Acc>
OFX_DEV HsOps hs_point_load(const Acc &acc, const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif, int i, int j,
                            int nx, int ny)
{
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    HsOps o;
    o.p1 = acc.get(iu, jl); o.p2 = acc.get(iu, jr);
    o.p3 = acc.get(id, jl); o.p4 = acc.get(id, jr);
    if (i == ny - 1 && j == nx - 1) {
        const double2 a1 = o.p1, a2 = o.p2;
        o.p1 = o.p3; o.p2 = o.p4; o.p3 = a1; o.p4 = a2;
    }
    o.p5 = acc.get(iu, j); o.p6 = acc.get(i, jl);
    o.p7 = acc.get(id, j); o.p8 = acc.get(i, jr);
    o.c = acc.get(i, j);
    acc.coef(A, Dif, i, j, o.a, o.dif);
    return o;
}
template <typename T, class Acc>
OFX_DEV double hs_point_finish(const Acc &acc, const HsOps &o, int i, int j, double alpha2)
{
    const double2 a = o.a;
    const double dif = o.dif;
    const double w = HS_SOR_W;
    const double Au = dif * a.x, Av = dif * a.y;                              // :133-134
    const double Du = a.x * a.x + alpha2, Dv = a.y * a.y + alpha2;            // :135-136
    const double D = a.x * a.y;                                               // :137
    const double ula = 1. / 12. * (o.p1.x + o.p2.x + o.p3.x + o.p4.x) + 1. / 6. * (o.p5.x + o.p6.x + o.p7.x + o.p8.x);
    const double vla = 1. / 12. * (o.p1.y + o.p2.y + o.p3.y + o.p4.y) + 1. / 6. * (o.p5.y + o.p6.y + o.p7.y + o.p8.y);
    const double uk = o.c.x, vk = o.c.y;
    const double un = rnd_to<T>((1.0 - w) * uk + w * (Au - D * vk + alpha2 * ula) / Du);   // :66
    const double vn = rnd_to<T>((1.0 - w) * vk + w * (Av - D * un + alpha2 * vla) / Dv);   // :67
    acc.put(i, j, make_double2(un, vn));
    return (un - uk) * (un - uk) + (vn - vk) * (vn - vk);                     // :70
}
template <typename T, class Acc>
OFX_DEV double hs_point_acc(const Acc &acc, const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif, int i,
                            int j, int nx, int ny, double alpha2)
{
    const HsOps o = hs_point_load<T>(acc, A, Dif, i, j, nx, ny);
    return hs_point_finish<T>(acc, o, i, j, alpha2);
}

template <typename T, bool COH = false, bool SNAP = false>
OFX_DEV double hs_point(typename Pix<T>::v2 *U, const typename Pix<T>::v2 *__restrict__ A,
                        const T *__restrict__ Dif, int i, int j, int nx, int ny, double alpha2,
                        typename Pix<T>::v2 *snap = nullptr)
{
    const UGlobal<T, COH, SNAP, LayRow> acc = {U, snap, LayRow{nx}};
    return hs_point_acc<T>(acc, A, Dif, i, j, nx, ny, alpha2);
}
// the same on hyperplane-major arrays (windowed sweeps)
template <typename T, bool COH, bool SNAP>
OFX_DEV double hs_point_skew(typename Pix<T>::v2 *U, const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif,
                             int i, int j, int nx, int ny, double alpha2, typename Pix<T>::v2 *snap)
{
    const UGlobal<T, COH, SNAP, LaySkew> acc = {U, snap, LaySkew{ny, HS_PLANE_C_SKEW}};
    return hs_point_acc<T>(acc, A, Dif, i, j, nx, ny, alpha2);
}

// One colour of one SOR sweep (fast, order-changing mode).
template <typename T>
__global__ __launch_bounds__(256) void k_hs_sor(typename Pix<T>::v2 *__restrict__ U,
                                                const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif,
                                                double *__restrict__ err, int k, int nx, int ny, int ci, int cj,
                                                double alpha2, double tol)
{
    const double prev = loop_fetch_prev(err, k);
    const int j = 2 * (blockIdx.x * 64 + threadIdx.x) + cj;
    const int i = 2 * (blockIdx.y * 4 + threadIdx.y) + ci;
    const bool in = (j < nx) && (i < ny);
    const int gw = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.y;
    if (!loop_continues(prev, k, nx * ny, tol, OFX_CRIT_SQRT_MEAN)) return;
    double e = 0.0;
    if (in) e = hs_point<T>(U, A, Dif, i, j, nx, ny, alpha2);
    loop_accumulate(err, k, e, gw);
}

// ---- exact mode: the reference's sequential sweep order, pipelined over hyperplanes --------------------
// Pixel X of sweep s runs at time t = pos(X) + C s.  pos() is the pixel's rank in a time-skewed version of
// the reference's visiting order (interior rows lexicographic -> first/last row -> first/last column ->
// corners, horn_schunck_pyramidal.cpp:148-228) and C the spacing between consecutive sweeps, chosen so
// that for every two NEIGHBOURING pixels X before Y in that order   t(X,s) < t(Y,s) < t(X,s+1).
// Then (a) all pixels of one time step are mutually independent, (b) the in-place array holds, at the
// moment X of sweep s runs, exactly the versions the sequential sweep would read (Y of sweep s if Y comes
// earlier, of sweep s-1 if later).  One launch = one time step, all sweeps of the batch that are inside
// the image at that time (blockIdx.y = sweep).  HS (8 neighbours): pos = 2i + j in the interior, C = 6.
// The same schedule is executed on the CPU by oracle order 2 and is bit-identical to the reference's
// sequential sweeps (tests/test_oracle_golden.py::test_hyperplane_schedule_is_exact).
#define HS_PLANE_C 6
OFX_DEV bool hs_plane_item(int r, int q, int nx, int ny, int corner, int &i, int &j)
{
    if (r >= 1 && r <= ny - 2) { i = r; j = q - 2 * r; return j >= 1 && j <= nx - 2; }          // interior
    if (r == 0) { i = 0; j = q - 4; return j >= 1 && j <= nx - 2; }                               // first row
    if (r == ny - 1) { i = ny - 1; j = q - 2 * (ny - 1); return j >= 1 && j <= nx - 2; }         // last row
    if (r == ny) { const int d = q - 4; i = d >> 1; j = 0; return !(d & 1) && i >= 1 && i <= ny - 2; }             // first column
    if (r == ny + 1) { const int d = q - nx - 1; i = d >> 1; j = nx - 1; return !(d & 1) && i >= 1 && i <= ny - 2; }  // last column
    if (r == ny + 2) {                                                                            // corners
        if (corner == 0) { i = 0; j = 0; return q == 7; }
        if (corner == 1) { i = 0; j = nx - 1; return q == nx + 4; }
        if (corner == 2) { i = ny - 1; j = 0; return q == 2 * ny + 1; }
        i = ny - 1; j = nx - 1; return q == 2 * ny + nx - 2;
    }
    return false;
}

template <typename T>
__global__ __launch_bounds__(64) void k_hs_plane(typename Pix<T>::v2 *__restrict__ U,
                                                 const typename Pix<T>::v2 *__restrict__ A, const T *__restrict__ Dif,
                                                 double *__restrict__ err, int t, int s_lo, int nx, int ny, double alpha2)
{
    const int s = s_lo + blockIdx.y;
    const int q = t - HS_PLANE_C * s;
    const int r = blockIdx.x * 64 + threadIdx.x;
    double e = 0.0;
    int i, j;
    if (r == ny + 2) {
        for (int corner = 0; corner < 4; corner++)
            if (hs_plane_item(r, q, nx, ny, corner, i, j)) e += hs_point<T>(U, A, Dif, i, j, nx, ny, alpha2);
    } else if (hs_plane_item(r, q, nx, ny, 0, i, j)) {
        e = hs_point<T>(U, A, Dif, i, j, nx, ny, alpha2);
    }
    loop_accumulate(err, s, e, blockIdx.x);
}

// ---- exact mode, windowed: K time steps per launch ----------------------------------------------------------
// One launch per time step makes the exact mode host-launch-bound (~4.5 us per step, thousands of steps per
// solve).  The schedule t = pos + C s stays valid for any LARGER sweep spacing, so sweeps are spaced K + C steps
// apart and a launch executes the K steps [tau0, tau0 + K) of every sweep in flight, ONE WORKGROUP PER SWEEP:
// sweep s runs its local steps q = tau - (K + C) s, thread r of its workgroup owns plane item r (k_hs_plane's
// numbering), and consecutive steps are separated by a store drain + workgroup barrier.  Everything sweep s
// reads from sweep s - 1 (pos <= q + C - 1, i.e. tau' <= tau - K - 1) was written in an EARLIER launch, so
// workgroups never communicate inside a launch: no flags, no spinning, no placement assumptions -- only
// same-workgroup hand-offs (L1-bypassing loads, ldu2) and kernel boundaries.  Launches per batch of S sweeps:
// (qmax + (K + C) S) / K instead of qmax + C S.
// Every sweep also writes its values into its own snapshot plane: the state after sweep n is snapshot n,
// whatever later sweeps of the batch have overwritten in place, so a batch that ran past the stopping sweep
// needs no rollback and re-run.
// OFX_SOR_COH = 1: unknowns are read past the L1 (sc1).  0 (default): plain loads -- within a workgroup the L1 is
// coherent (same CU, write-through), and everything read from other sweeps was written before this launch began.
#ifndef OFX_SOR_COH
#define OFX_SOR_COH 0
#endif
// Geometry of one windowed launch (see sor_window_loop).  A sweep is further cut into row blocks of R rows,
// one workgroup per (sweep, block): block b runs lag_b steps behind block b - 1 of its sweep, sweeps are
// lag_s apart, and a border pixel is executed by the block of the interior row it depends on last.
struct SorWin {
    int tau0, K, lag_s, lag_b, lag_f, s_first, R;      // lag_f: spacing of the frames of a sequence (temporal Brox)
};
// Lockstep group: G image pairs solved by the same launches (blockIdx.z = pair), every array of a level holding the
// pairs back to back.  A lone solve is a latency chain of qmax + lag_s * sweeps dependent steps with a few hundred
// waves of work each; the pairs of a group share that chain.  Each pair keeps its own error slots, snapshots and
// stopping test (bit g of runmask: pair g still sweeps in this solve), so its flow is the one it gets alone.
struct SorGrp {
    unsigned runmask;
    int      err_stride;     // doubles between the error slots of consecutive pairs
    size_t   npix;           // elements of one (hyperplane-major, padded) plane = distance between consecutive pairs
    size_t   snap_stride;    // elements between the snapshot planes of consecutive pairs
};
// which plane item thread t of block b plays: its R rows, then the three shared items (first column, last
// column, corners) of which a block accepts only the pixels assigned to it (sor_border_block)
OFX_DEV int sor_window_item(const SorWin &w, int b, int t, int ny)
{
    if (t < w.R) { const int r = b * w.R + t; return r < ny ? r : -1; }
    return t < w.R + 3 ? ny + (t - w.R) : -1;
}
// block of a border-column / corner pixel: the row below for the columns (the interior pixel (i+1, 1) precedes
// (i, 0) in the reference order), row 2 for the top corners, the last row for the bottom corners
OFX_DEV int sor_border_block(int i, int ny, int R)
{
    int r;
    if (i == 0) r = ny - 1 < 2 ? ny - 1 : 2;
    else if (i == ny - 1) r = ny - 1;
    else r = i + 1;
    return r / R;
}

// Local steps at which row block b has any pixel to update (rows b R .. min(b R + R, ny) - 1 and the border pixels assigned to
// it, sor_border_block): [c b R, c r_last + nx + 5] for the stencil skew c (2 Horn-Schunck, 1 Brox) -- half of the (sweep, block)
// units of a launch lie outside it and leave at once (tools/check_sor_schedule.py unit_range, tests/test_host_logic.py).
OFX_DEV bool sor_unit_idle(const SorWin &w, int b, int q_first, int K, int c, int nx, int ny)
{
    const int rf = b * w.R, rl = (rf + w.R < ny ? rf + w.R : ny) - 1;
    return q_first > c * rl + nx + 5 || q_first + K - 1 < c * rf;
}

// SPW consecutive sweeps of one row block share a workgroup: their updates are independent (lag_s steps apart), so their
// loads are issued together and ONE store drain + barrier per step serves all of them -- the per-step latency, not
// arithmetic or bandwidth, is what bounds a lockstep group (profiles/r02_f_sor_counters.txt).
template <typename T, int SPW, int MAXT>
__global__ __launch_bounds__(MAXT) void k_hs_window(typename Pix<T>::v2 *Ug, typename Pix<T>::v2 *snap,
                                                    const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Difg,
                                                    double *__restrict__ errg, SorWin w, SorGrp grp, int s_cnt, int nx, int ny,
                                                    double alpha2)
{
    const int b = blockIdx.x, s0 = w.s_first + blockIdx.y * SPW, g = blockIdx.z;
    if (!((grp.runmask >> g) & 1u)) return;                      // this pair's solve has already stopped
    const int qmax = 2 * ny + nx - 2;
    int q_first[SPW];
    bool live[SPW], any = false;
#pragma unroll
    for (int u = 0; u < SPW; u++) {
        q_first[u] = w.tau0 - w.lag_s * (s0 + u) - w.lag_b * b;
        live[u] = (s0 + u < w.s_first + s_cnt) && !sor_unit_idle(w, b, q_first[u], w.K, 2, nx, ny);
        any = any || live[u];
    }
    if (!any) return;                                            // none of these (sweep, block) units has a step in the window
    typename Pix<T>::v2 *U = Ug + g * grp.npix;
    const typename Pix<T>::v2 *__restrict__ A = Ag + g * grp.npix;
    const T *__restrict__ Dif = Difg + g * grp.npix;
    double *__restrict__ err = errg + (size_t) g * grp.err_stride;
    const int r = sor_window_item(w, b, threadIdx.x, ny);
    double e[SPW];
#pragma unroll
    for (int u = 0; u < SPW; u++) e[u] = 0.0;
    for (int k = 0; k < w.K; k++) {
        bool have[SPW];
        int pi[SPW], pj[SPW];
        HsOps ops[SPW];
#pragma unroll
        for (int u = 0; u < SPW; u++) {                          // all loads of this step first ...
            const int q = q_first[u] + k;
            have[u] = live[u] && q >= 0 && q <= qmax && r >= 0 && r != ny + 2 && hs_plane_item(r, q, nx, ny, 0, pi[u], pj[u]) &&
                      (r < ny || sor_border_block(pi[u], ny, w.R) == b);
            if (SPW > 1 && have[u]) {
                const UGlobal<T, OFX_SOR_COH != 0, true, LaySkew> acc = {U, snap + g * grp.snap_stride + (size_t) (s0 + u) * grp.npix,
                                                                         LaySkew{ny, HS_PLANE_C_SKEW}};
                ops[u] = hs_point_load<T>(acc, A, Dif, pi[u], pj[u], nx, ny);
            }
        }
#pragma unroll
        for (int u = 0; u < SPW; u++) {                          // ... then the updates
            typename Pix<T>::v2 *mysnap = snap + g * grp.snap_stride + (size_t) (s0 + u) * grp.npix;
            if (have[u]) {
                const UGlobal<T, OFX_SOR_COH != 0, true, LaySkew> acc = {U, mysnap, LaySkew{ny, HS_PLANE_C_SKEW}};
                // one sweep per workgroup: load and update in one go (the compiler interleaves them in 66 VGPRs = 7 waves
                // per SIMD; holding every operand first would take 106)
                if (SPW > 1) e[u] += hs_point_finish<T>(acc, ops[u], pi[u], pj[u], alpha2);
                else e[u] += hs_point_acc<T>(acc, A, Dif, pi[u], pj[u], nx, ny, alpha2);
            }
            const int q = q_first[u] + k;
            if (live[u] && r == ny + 2 && q >= 0 && q <= qmax) {  // the corner item: up to four pixels, one after the other
                int i, j;
                for (int corner = 0; corner < 4; corner++)
                    if (hs_plane_item(r, q, nx, ny, corner, i, j) && sor_border_block(i, ny, w.R) == b)
                        e[u] += hs_point_skew<T, OFX_SOR_COH != 0, true>(U, A, Dif, i, j, nx, ny, alpha2, mysnap);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this step's stores have reached L2 ...
        __syncthreads();                                       // ... before any wave of the workgroup reads them
    }
#pragma unroll
    for (int u = 0; u < SPW; u++)
        if (live[u]) loop_accumulate(err, s0 + u, e[u], b * 4 + (threadIdx.x >> 6));
}

// ---- windowed exact mode with the launch window staged in LDS ---------------------------------------------------------------
// k_hs_window pays one global round trip per time step: the step's stores have to reach L2 (s_waitcnt vmcnt(0)) before the
// barrier lets the next step load them -- ~1.5 us per step for ~200 instructions of work.  The schedule guarantees that
// everything a workgroup takes from ANOTHER workgroup was written in an earlier launch and is overwritten in a later one, so
// a workgroup can fetch everything its KW steps will touch when the launch starts -- all loads in flight together, one
// memory latency per launch -- and run the steps without global loads: the unknowns live in an LDS window (neighbours are
// LDS reads; an update goes to LDS for the workgroup's own later steps and, without waiting, to the in-place array and the
// sweep's snapshot for everybody else; the per-step barrier waits for LDS only), the constant operands of the KW pixels a
// thread will update wait in LDS slots of its own.  What the KW steps [q0, q0 + KW) of (sweep, row block b) touch
// (tools/check_sor_schedule.py --cover enumerates it, tests/test_host_logic.py checks it): unknowns on hyperplanes
// q0 - 7 .. q0 + KW + 2 of the skewed coordinate c i + j (the border pixels run up to 7 steps behind their own coordinate),
// rows b R - 2 .. b R + R.  Same per-pixel function, same operands: bit-identical to k_hs_window.
template <typename T, int KW> struct HsWinLds {
    static constexpr int NH = KW + 10;          // hyperplanes of unknowns in the window
    static constexpr int BACK = 7;              // window starts at q0 - BACK
};
template <typename T, bool SNAP> struct ULds {
    typename Pix<T>::v2 *U, *snap;               // global arrays (hyperplane-major), written through
    double2 *win;                                // [NH][nr]  unknowns
    int nr, row0, h0;                            // LDS rows, image row of LDS row 0, hyperplane of LDS plane 0
    LaySkew lay;
    double2 a;                                   // constant operands of the pixel being updated (set per step)
    double dif;
    OFX_DEV double2 get(int ii, int jj) const { return win[(2 * ii + jj - h0) * nr + (ii - row0)]; }
    OFX_DEV void coef(const typename Pix<T>::v2 *, const T *, int, int, double2 &a_, double &dif_) const { a_ = a; dif_ = dif; }
    OFX_DEV void put(int i, int j, double2 v) const
    {
        win[(2 * i + j - h0) * nr + (i - row0)] = v;
        const size_t p = lay.idx(i, j);
        stn2(U + p, v);
        if (SNAP) stn2(snap + p, v);
    }
};
// the pixel plane item r updates at local step q in row block b, if any (the rule of k_hs_window)
OFX_DEV bool hs_window_pixel(const SorWin &w, int b, int r, int q, int qmax, int nx, int ny, int &pi, int &pj)
{
    return q >= 0 && q <= qmax && r >= 0 && r != ny + 2 && hs_plane_item(r, q, nx, ny, 0, pi, pj) &&
           (r < ny || sor_border_block(pi, ny, w.R) == b);
}
template <typename T, int KW, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_hs_window_lds(
    typename Pix<T>::v2 *Ug, typename Pix<T>::v2 *snap, const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Difg,
    double *__restrict__ errg, SorWin w, SorGrp grp, int s_cnt, int nx, int ny, double alpha2)
{
    using W = HsWinLds<T, KW>;
    extern __shared__ double2 hs_lds[];
    const int b = blockIdx.x, s = w.s_first + blockIdx.y, g = blockIdx.z;
    if (!((grp.runmask >> g) & 1u)) return;                      // this pair's solve has already stopped
    const int qmax = 2 * ny + nx - 2;
    const int q0 = w.tau0 - w.lag_s * s - w.lag_b * b;
    if (sor_unit_idle(w, b, q0, KW, 2, nx, ny)) return;          // no step of this (sweep, block) in the window: uniform
    typename Pix<T>::v2 *U = Ug + g * grp.npix;
    const typename Pix<T>::v2 *__restrict__ A = Ag + g * grp.npix;
    const T *__restrict__ Dif = Difg + g * grp.npix;
    typename Pix<T>::v2 *mysnap = snap + g * grp.snap_stride + (size_t) s * grp.npix;
    double *__restrict__ err = errg + (size_t) g * grp.err_stride;
    const int nr = w.R + 3, row0 = b * w.R - 2, h0 = q0 - W::BACK, t = threadIdx.x;
    const int nt = blockDim.x;
    double2 *win = hs_lds;                                       // [NH][nr] unknowns
    double2 *ca = hs_lds + W::NH * nr;                           // [KW][nt] (I2wx, I2wy) of the pixel thread t updates at step k
    double *cd = reinterpret_cast<double *>(ca + KW * nt);       // [KW][nt] dif
    const LaySkew lay = {ny, HS_PLANE_C_SKEW};
    const int r = sor_window_item(w, b, t, ny);
    // ---- fetch: thread t brings LDS row t (image row row0 + t) of every hyperplane of the window and the constant operands of
    // the KW pixels it will update; every load unconditional (entries outside the image / steps without a pixel read a
    // harmless address and are never used) and issued before the first use: one memory latency for the whole launch
    double2 u_in[W::NH], a_in[KW];
    double d_in[KW];
    {
        const int i = row0 + t, ic = i < 0 ? 0 : (i > ny - 1 ? ny - 1 : i);
        const int hmax = 2 * (ny - 1) + nx - 1;
#pragma unroll
        for (int k = 0; k < W::NH; k++) {
            const int h = h0 + k, hc = h < 0 ? 0 : (h > hmax ? hmax : h);
            u_in[k] = ldw2(U + (size_t) hc * ny + ic);
        }
#pragma unroll
        for (int k = 0; k < KW; k++) {
            int pi, pj;
            const bool have = hs_window_pixel(w, b, r, q0 + k, qmax, nx, ny, pi, pj);
            const size_t p = have ? lay.idx(pi, pj) : 0;
            a_in[k] = ldw2(A + p);
            d_in[k] = ldw(Dif + p);
        }
        if (t < nr) {
#pragma unroll
            for (int k = 0; k < W::NH; k++) win[k * nr + t] = u_in[k];
        }
#pragma unroll
        for (int k = 0; k < KW; k++) { ca[k * nt + t] = a_in[k]; cd[k * nt + t] = d_in[k]; }     // this thread's own slots
    }
    __syncthreads();
    ULds<T, true> acc = {U, mysnap, win, nr, row0, h0, lay, make_double2(0.0, 0.0), 0.0};
    double e = 0.0;
#pragma unroll 1
    for (int k = 0; k < KW; k++) {
        const int q = q0 + k;
        int pi, pj;
        if (hs_window_pixel(w, b, r, q, qmax, nx, ny, pi, pj)) {
            acc.a = ca[k * nt + t];
            acc.dif = cd[k * nt + t];
            e += hs_point_acc<T>(acc, A, Dif, pi, pj, nx, ny, alpha2);
        }
        if (q >= 0 && q <= qmax && r == ny + 2) {                // the corner item: up to four pixels, one after the other
            for (int corner = 0; corner < 4; corner++)
                if (hs_plane_item(r, q, nx, ny, corner, pi, pj) && sor_border_block(pi, ny, w.R) == b) {
                    const size_t p = lay.idx(pi, pj);            // four pixels per sweep: their operands come from memory
                    acc.a = ldw2(A + p);
                    acc.dif = ldw(Dif + p);
                    e += hs_point_acc<T>(acc, A, Dif, pi, pj, nx, ny, alpha2);
                }
        }
        // the next step reads this step's LDS writes: wait for LDS only (the global stores drain on their own)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    loop_accumulate(err, s, e, b * 4 + (t >> 6));
}

// Batch driver of the windowed exact mode.  launch(w, blocks, sweeps) enqueues one window over `blocks` row
// blocks x `sweeps` sweeps starting at w.s_first; take(n) makes snapshot n - 1 the current state.  Same contract
// as sor_exact_loop.  Spacing: lag_b = K between the row blocks of a sweep, lag_s = 2 K + C between sweeps (K + C
// when a sweep is one block): with these every value a workgroup reads from ANOTHER (sweep, block) was written
// at least one launch earlier and is overwritten at least one launch later (checked exhaustively on small
// images for both stencils, tools/check_sor_schedule.py).
// nz > 1 (temporal Brox): a sweep visits nz frames one after the other; frame number o (in visiting order) runs
// lag_f = K steps behind frame o - 1 and the sweeps move lag_f (nz - 1) further apart.
template <class WindowFn, class TakeFn>
static int sor_window_loop(ofx_ctx *ctx, int G, int size, int ny, double TOL, int maxiter, int qmax, int C, int batch,
                           WindowFn launch, TakeFn take, int *n_out, double *err_out, int nz = 1, int *hint = nullptr,
                           int Kdef = 0)          // Kdef: the caller's default steps per launch (0 = the rule below)
{
    // G problems in lockstep (SorGrp): launch(w, blocks, sweeps, runmask, err_stride) serves every problem whose bit
    // is set; take(g, n) makes snapshot n - 1 of problem g its current state.  All problems start together, so the
    // ones still sweeping have all run the same number of full batches.
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "sor group of %d problems", G);
    // Measured on MI355X with the hyperplane-major layout (profiles/r02_e_sor_geometry_sweep.jsonl): a lone solve wants
    // many small workgroups (64 rows: more row blocks in flight along the latency chain), a lockstep group has enough
    // workgroups anyway and runs 6-10 % faster with 125 rows (+ the 3 border items = two full waves); 8 steps per launch,
    // 4 for the 4-neighbour stencil in a group (its sweeps are spaced C = 2 apart, so the 2 K lag dominates the pipeline).
    SorWin w;
    w.K = ctx->sor_window > 0 ? ctx->sor_window : (Kdef > 0 ? Kdef : ((G >= 4 && C <= 2) ? 4 : 8));
    w.R = ctx->sor_rows > 0 ? ctx->sor_rows : (G >= 4 ? 125 : 64);
    if (w.R < 2) w.R = 2;
    if (w.R > 1021) w.R = 1021;                                  // R + 3 threads per workgroup
    const int B = ofx_cdiv(ny, w.R);
    w.lag_b = w.K;
    w.lag_f = nz > 1 ? w.K : 0;
    w.lag_s = (B > 1 ? 2 * w.K : w.K) + C + w.lag_f * (nz - 1);
    const int per = batch + 1;                                   // error slots per problem
    OFX_TRY(ofx_loop_reserve(ctx, G * per));
    LoopSpec LS;
    LS.size = size;
    LS.thr = TOL;
    LS.crit = OFX_CRIT_SQRT_MEAN;
    LS.chunk = 0;
    LS.fixed = false;
    LS.pairs = false;
    // Every sweep of a batch costs lag_s steps of pipeline whether it is needed or not, so batches are sized, not
    // maximal (`batch` is the snapshot capacity): the first one from the sweep count of the previous solve at this level
    // (*hint = the largest of the group; consecutive warps / outer iterations converge in similar, usually decreasing,
    // numbers of sweeps) with 25 % + 2 of head room, 32 without a hint; a solve that needs more runs further,
    // half-sized batches.
    int first = batch < 32 ? batch : 32;
    if (ctx->sor_batch > 0) first = batch;                       // explicit option: fixed batches
    else if (hint && *hint > 0) {
        first = *hint + *hint / 4 + 2;
        first = first < 8 ? 8 : (first > batch ? batch : first);
    }
    const int later = ctx->sor_batch > 0 ? batch : (first / 2 < 8 ? (batch < 8 ? batch : 8) : first / 2);
    int niter = 0;                                               // sweeps of the problems that are still active
    unsigned active = (1000 > TOL && maxiter > 0) ? ((G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u)) : 0u;
    for (int g = 0; g < G; g++) { n_out[g] = 0; err_out[g] = 1000; }                       // :140 / :312
    while (active) {
        const int want = niter == 0 ? first : later;
        const int ns = (maxiter - niter < want) ? maxiter - niter : want;
        OFX_TRY(ofx_loop_clear(ctx, (size_t) G * per));
        const long tail = (long) w.lag_b * (B - 1) + (long) w.lag_f * (nz - 1);   // the last unit of a sweep ends this much later
        const long total = (long) qmax + 1 + tail + (long) w.lag_s * (ns - 1);
        for (long tau0 = 0; tau0 < total; tau0 += w.K) {
            long s_lo = (tau0 - qmax - tail + w.lag_s - 1) / w.lag_s;   // smallest s still inside the image
            if (tau0 - qmax - tail < 0) s_lo = 0;
            long s_hi = (tau0 + w.K - 1) / w.lag_s;              // largest s that has started
            if (s_hi > ns - 1) s_hi = ns - 1;
            if (s_hi < s_lo) continue;
            w.tau0 = (int) tau0;
            w.s_first = (int) s_lo;
            OFX_TRY(launch(w, B, (int) (s_hi - s_lo + 1), active, per * OFX_NSHARD));
        }
        LS.max_iter = ns;
        const int seq = ofx_poll_seq(ctx);
        const int slot = (int) (ctx->poll_seq++ % OFX_NPOLL);
        OFX_TRY(ofx_loop_finalize_group(ctx, LS, G, per, 0, ns, &ctx->h_state[slot * OFX_MAX_GROUP], seq));
        OFX_HIP(ctx, hipEventRecord(ctx->ev_poll[slot], ctx->stream));
        OFX_HIP(ctx, hipEventSynchronize(ctx->ev_poll[slot]));
        for (int g = 0; g < G; g++) {
            if (!((active >> g) & 1u)) continue;
            const OfxIterState st = ctx->h_state[slot * OFX_MAX_GROUP + g];
            if (st.n < ns) OFX_TRY(take(g, st.n));              // stopped inside the batch: the state is snapshot n - 1
            n_out[g] = niter + st.n;
            err_out[g] = st.error;
            if (!(st.error > TOL && n_out[g] < maxiter)) active &= ~(1u << g);
        }
        niter += ns;
    }
    if (hint) {
        int h = 0;
        for (int g = 0; g < G; g++) h = n_out[g] > h ? n_out[g] : h;
        *hint = h;
    }
    return OFX_OK;
}

// Exact SOR loop:  while (error > TOL && n < maxiter) sweep.   Sweeps run in batches of up to `batch`
// pipelined sweeps (launch_plane(t, s_lo, s_cnt) = one time step).  The per-sweep errors of a batch are
// only known when it has drained, so a batch that ran past the stopping sweep is rolled back to its
// checkpoint (a copy of the unknowns taken before the batch) and re-run with exactly the sweeps that
// count.  qmax = largest pos() of the image, C = sweep spacing.
template <class PlaneFn, class SaveFn, class RestoreFn>
static int sor_exact_loop(ofx_ctx *ctx, int size, double TOL, int maxiter, int qmax, int C, PlaneFn launch_plane,
                          SaveFn save, RestoreFn restore, int *n_out, double *err_out)
{
    int niter = 0;
    double error = 1000;
    const int batch = ctx->sor_batch > 0 ? ctx->sor_batch : 64;
    OFX_TRY(ofx_loop_reserve(ctx, batch + 1));
    LoopSpec LS;
    LS.size = size;
    LS.thr = TOL;
    LS.crit = OFX_CRIT_SQRT_MEAN;
    LS.chunk = 0;
    LS.fixed = false;
    LS.pairs = false;
    auto run_batch = [&](int ns, OfxIterState *out) -> int {
        OFX_TRY(ofx_loop_clear(ctx, (size_t) ns));
        const int tmax = qmax + C * (ns - 1);
        for (int t = 0; t <= tmax; t++) {
            int s_lo = (t - qmax + C - 1) / C;                 // smallest s with t - C s <= qmax
            if (t - qmax < 0) s_lo = 0;
            int s_hi = t / C;
            if (s_hi > ns - 1) s_hi = ns - 1;
            if (s_hi < s_lo) continue;
            OFX_TRY(launch_plane(t, s_lo, s_hi - s_lo + 1));
        }
        LS.max_iter = ns;
        const int slot = (int) (ctx->poll_seq++ % OFX_NPOLL);
        OFX_TRY(ofx_loop_finalize_group(ctx, LS, 1, 0, 0, ns, &ctx->h_state[slot * OFX_MAX_GROUP], ofx_poll_seq(ctx)));
        OFX_HIP(ctx, hipEventRecord(ctx->ev_poll[slot], ctx->stream));
        OFX_HIP(ctx, hipEventSynchronize(ctx->ev_poll[slot]));
        *out = ctx->h_state[slot * OFX_MAX_GROUP];
        return OFX_OK;
    };
    while (error > TOL && niter < maxiter) {
        const int b = (maxiter - niter < batch) ? maxiter - niter : batch;
        OFX_TRY(save());
        OfxIterState st;
        OFX_TRY(run_batch(b, &st));
        if (st.n < b) {                                        // stopped inside the batch: redo exactly st.n sweeps
            const int used = st.n;
            OFX_TRY(restore());
            OFX_TRY(run_batch(used, &st));
            st.n = used;
        }
        niter += st.n;
        error = st.error;
    }
    *n_out = niter;
    *err_out = error;
    return OFX_OK;
}

// snapshot capacity (= most sweeps in one batch) of the windowed mode: option "sor_batch", else 64, capped by maxiter,
// by ~4 GiB of snapshots per pair and -- for all pairs of a lockstep group -- by a third of the device memory that is
// free right now, shared between the contexts expected to solve concurrently (option "concurrency"; the batch entry
// points set it to their number of contexts)
static int sor_pick_batch(const ofx_ctx *ctx, size_t npix, size_t elem_bytes, int maxiter, int G = 1)
{
    int b = ctx->sor_batch > 0 ? ctx->sor_batch : 64;
    const double cap = (double) ((size_t) 4 << 30);
    double cap_group = (double) ((size_t) 40 << 30);
    size_t mfree = 0, mtotal = 0;
    if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
        const double share = 0.33 * (double) mfree / (ctx->concurrency > 1 ? ctx->concurrency : 1);
        if (share < cap_group) cap_group = share;
    } else {
        (void) hipGetLastError();
    }
    const double plane = (double) npix * (double) elem_bytes;
    while (b > 8 && (b * plane > cap || b * plane * G > cap_group)) b /= 2;
    while (b > 1 && b * plane > cap) b /= 2;
    if (b > maxiter) b = maxiter;
    return b < 1 ? 1 : b;
}
// sweeps per workgroup of the windowed kernels: option "sor_spw" (1, 2 or 4), default 1.  Measured with 16 pairs in
// lockstep (profiles/r02_i_sor_sweeps_per_workgroup.jsonl): 2 / 4 sweeps per workgroup -- their loads issued together,
// one store drain + barrier per step for all of them -- run at 0.66x / 0.47x the throughput of 1 (HS 23.9k -> 15.7k ->
// 11.2k Mpix*sweeps/s): the operands of two updates take 151 VGPRs = 3 waves per SIMD instead of 8 at 64, and the
// per-step latency is hidden by resident waves, not by instruction-level parallelism inside one.
static int sor_pick_spw(const ofx_ctx *ctx, int G)
{
    (void) G;
    if (ctx->sor_spw == 1 || ctx->sor_spw == 2 || ctx->sor_spw == 4) return ctx->sor_spw;
    return 1;
}
// Which windowed kernel: option "sor_lds" = 0 the per-step global round trip (k_*_window), 2 the LDS-staged launch window
// (k_*_window_lds), 1 (default) by measurement: the LDS window for a lone solve -- a latency chain, where one memory latency
// per launch instead of one per step counts -- and the global kernels for lockstep groups of 4 pairs and more, which are
// bound by resident waves (28 per CU at 64 VGPRs and no LDS against a few LDS-limited workgroups).
// Measured (profiles/r03_e_sor_window_kernels_lds_vs_global.txt, 16 P1 pairs of cfg 3 / cfg 4): Horn-Schunck lone solves
// 572 -> 537 ms with 8 steps per launch (16: 617), lockstep groups of 16 81 -> 147 ms per pair (116 since idle units leave at
// once, profiles/r03_n_hs_window_kernels_counters.txt: 800 resident waves against 2400); Brox lone solves 255 -> 297 ms.
// A step is bound by the ~200 dependent instructions of a lone wave, not by the round trip the window removes.
static bool sor_use_lds(const ofx_ctx *ctx, int G, bool brox = false)
{
    return ctx->sor_lds == 2 || (ctx->sor_lds == 1 && G < 4 && !brox);
}
static int sor_window_threads(int n_items)
{
    const int t = ofx_cdiv(n_items, 64) * 64;
    return t > 1024 ? 1024 : t;
}

// One level of a lockstep group: every array holds G pairs back to back (pair g at element g * nx * ny).
template <typename T> struct HsLevel {
    int nx, ny, G;
    T *I1, *I2;
    typename Pix<T>::v2 *pa;    // (I2, I2x)
    T *pb;                      // I2y
    typename Pix<T>::v2 *U, *A, *Uck;
    T *Dif;
    // windowed exact mode, hyperplane-major (LaySkew) and allocated on first use: the unknowns, the coefficients and
    // snap_planes snapshot planes per pair
    typename Pix<T>::v2 *Us, *As, *Snap;
    T *Difs;
    int snap_planes;
    int sweep_hint;             // sweeps of the previous solve at this level (sizes the next first batch)
    unsigned cur;               // tile sweeps (sor_exact = 0): bit g set = the flow of pair g lives in Uck, not U
    size_t n() const { return (size_t) nx * ny; }
    typename Pix<T>::v2 *flow(int g) const { return (((cur >> g) & 1u) ? Uck : U) + (size_t) g * n(); }
};

template <typename T> static int hs_level_alloc(ofx_ctx *ctx, HsLevel<T> &L, int nx, int ny, int G, bool images)
{
    const size_t n = (size_t) nx * ny * G;
    L.nx = nx;
    L.ny = ny;
    L.G = G;
    L.I1 = L.I2 = nullptr;
    if (images) {
        OFX_TRY(ofx_alloc(ctx, n, &L.I1));
        OFX_TRY(ofx_alloc(ctx, n, &L.I2));
    }
    OFX_TRY(ofx_alloc(ctx, n, &L.pa));
    OFX_TRY(ofx_alloc(ctx, n, &L.pb));
    OFX_TRY(ofx_alloc(ctx, n, &L.U));
    OFX_TRY(ofx_alloc(ctx, n, &L.Uck));
    OFX_TRY(ofx_alloc(ctx, n, &L.A));
    OFX_TRY(ofx_alloc(ctx, n, &L.Dif));
    L.Snap = L.Us = L.As = nullptr;
    L.Difs = nullptr;
    L.snap_planes = 0;
    L.sweep_hint = 0;
    L.cur = 0;
    return OFX_OK;
}

struct HsParams {
    double alpha, TOL;
    int warps, maxiter, verbose;
};

// src/horn_schunck_pyramidal.cpp:78-249 on device data for the G pairs of a lockstep group; L.U holds the incoming
// flows; stats[g] = work record of pair g.
template <typename T>
static int hs_single_scale_dev(ofx_ctx *ctx, HsLevel<T> &L, const HsParams &P, int scale, ofx_stats *stats)
{
    const int nx = L.nx, ny = L.ny, G = L.G;
    const size_t npix = L.n();
    const double alpha2 = P.alpha * P.alpha;
    if (P.verbose && G == 1)
        fprintf(stderr, "Single-scale Horn-Schunck of a %dx%d image\n\ta=%g nw=%d eps=%g mi=%d v=%d\n", nx, ny, P.alpha,
                P.warps, P.TOL, P.maxiter, P.verbose);
    const bool windowed = ctx->sor_exact == 1 && nx >= 3 && ny >= 3;
    const bool tiled = ctx->sor_exact == 0 && ctx->sor_fuse >= 0;        // colour order, K sweeps per launch (ofx_sor_tile.hip)
    if (G > 1 && !windowed && !tiled)
        return ofx_fail(ctx, OFX_ERR_ARG, "hs: lockstep groups need sor_exact = 1 (levels of at least 3x3, this one %dx%d) or the tile sweeps of sor_exact = 0", nx, ny);
    for (int g = 0; g < G; g++)
        OFX_TRY(op_grad_pack<T>(ctx, L.I2 + g * npix, L.pa + g * npix, L.pb + g * npix, nx, ny));      // :114
    const dim3 gw(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G);
    const dim3 gc(ofx_cdiv(ofx_cdiv(nx, 2), 64), ofx_cdiv(ofx_cdiv(ny, 2), 4));
    for (int w = 0; w < P.warps; w++) {
        if (P.verbose && G == 1) fprintf(stderr, "Warping %d:", w);
        hipLaunchKernelGGL(k_hs_warp<T>, gw, b2d(), 0, ctx->stream, L.pa, (const T *) L.pb, (const T *) L.I1, L.U, L.Uck, L.cur,
                           L.A, L.Dif, nx, ny);                                               // :123-137
        OFX_LAUNCH_CHECK(ctx);
        int niter[OFX_MAX_GROUP] = {0};
        double error[OFX_MAX_GROUP];
        for (int g = 0; g < G; g++) error[g] = 1000;                                          // :140
        float ms = 0.f;
        if (tiled) {
            OFX_TRY(ofx_hs_tile_solve<T>(ctx, G, L.U, L.Uck, &L.cur, L.A, (const T *) L.Dif, nx, ny, alpha2, P.TOL, P.maxiter,
                                         ctx->sor_fuse, niter, error, ctx->profile ? &ms : nullptr));
        } else if (windowed) {
            // windowed exact mode (default): the sweeps run on hyperplane-major copies of U, A, Dif
            const size_t ps = skew_plane_elems(nx, ny, HS_PLANE_C_SKEW);
            const size_t ub = ps * sizeof(typename Pix<T>::v2);
            const int batch = sor_pick_batch(ctx, ps, sizeof(typename Pix<T>::v2), P.maxiter, G);
            if (!L.Us) {
                OFX_TRY(ofx_alloc(ctx, ps * G, &L.Us));
                OFX_TRY(ofx_alloc(ctx, ps * G, &L.As));
                OFX_TRY(ofx_alloc(ctx, ps * G, &L.Difs));
            }
            if (L.snap_planes < batch) {
                OFX_TRY(ofx_alloc(ctx, ps * batch * G, &L.Snap));
                L.snap_planes = batch;
            }
            OFX_TRY((op_skew<typename Pix<T>::v2, true>(ctx, L.U, L.Us, nx, ny, HS_PLANE_C_SKEW, G)));
            OFX_TRY((op_skew<typename Pix<T>::v2, true>(ctx, L.A, L.As, nx, ny, HS_PLANE_C_SKEW, G)));
            OFX_TRY((op_skew<T, true>(ctx, L.Dif, L.Difs, nx, ny, HS_PLANE_C_SKEW, G)));
            const size_t snap_stride = ps * L.snap_planes;
            auto window = [&](const SorWin &w, int blocks, int sweeps, unsigned runmask, int err_stride) -> int {
                const SorGrp grp = {runmask, err_stride, ps, snap_stride};
                const int spw = sor_pick_spw(ctx, G);
                const size_t lds_need = (size_t) (w.R + 3) * (w.K + 10) * sizeof(double2) +
                                        (size_t) sor_window_threads(w.R + 3) * w.K * (sizeof(double2) + sizeof(double));
                if (sor_use_lds(ctx, G) && (w.K == 8 || w.K == 16 || w.K == 24) && w.R + 3 <= 256 && lds_need <= 160 * 1024) {
                    // the launch window staged in LDS (k_hs_window_lds): one sweep per workgroup, steps on LDS only
                    const dim3 grid(blocks, sweeps, G), blk(sor_window_threads(w.R + 3));
                    const int nr = w.R + 3;
#define OFX_HS_WINL(K_)                                                                                                        \
    do {                                                                                                                       \
        const size_t lds = (size_t) nr * HsWinLds<T, K_>::NH * sizeof(double2) + (size_t) blk.x * K_ * (sizeof(double2) + sizeof(double)); \
        if (lds > 160 * 1024) return ofx_fail(ctx, OFX_ERR_ARG, "hs: window of %d steps x %d rows does not fit the LDS", K_, w.R); \
        if (lds > 64 * 1024)                                                                                                   \
            OFX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hs_window_lds<T, K_, 256>),                      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                          \
        hipLaunchKernelGGL((k_hs_window_lds<T, K_, 256>), grid, blk, lds, ctx->stream, L.Us, L.Snap, L.As, (const T *) L.Difs, \
                           ctx->d_err, w, grp, sweeps, nx, ny, alpha2);                                                        \
    } while (0)
                    if (w.K == 8) OFX_HS_WINL(8);
                    else if (w.K == 16) OFX_HS_WINL(16);
                    else OFX_HS_WINL(24);
#undef OFX_HS_WINL
                    OFX_LAUNCH_CHECK(ctx);
                    return OFX_OK;
                }
                const dim3 grid(blocks, ofx_cdiv(sweeps, spw), G), blk(sor_window_threads(w.R + 3));
                // workgroups of up to 128 threads (the default geometries) are compiled without the 128-VGPR cap that a
                // 1024-thread bound implies: two sweeps per workgroup keep their operands in registers instead of spilling
#define OFX_HS_WIN(SPW_, MAXT_)                                                                                          \
    hipLaunchKernelGGL((k_hs_window<T, SPW_, MAXT_>), grid, blk, 0, ctx->stream, L.Us, L.Snap, L.As, (const T *) L.Difs, \
                       ctx->d_err, w, grp, sweeps, nx, ny, alpha2)
                if (blk.x <= 128 && spw > 1) {
                    if (spw == 4) OFX_HS_WIN(4, 128);
                    else OFX_HS_WIN(2, 128);
                } else {                                         // one sweep per workgroup: the 64-VGPR build, 8 waves per SIMD
                    if (spw == 4) OFX_HS_WIN(4, 1024);
                    else if (spw == 2) OFX_HS_WIN(2, 1024);
                    else OFX_HS_WIN(1, 1024);
                }
#undef OFX_HS_WIN
                OFX_LAUNCH_CHECK(ctx);
                return OFX_OK;
            };
            auto take = [&](int g, int n) -> int {
                OFX_HIP(ctx, hipMemcpyAsync(L.Us + g * ps, L.Snap + g * snap_stride + (size_t) (n - 1) * ps, ub,
                                            hipMemcpyDeviceToDevice, ctx->stream));
                return OFX_OK;
            };
            OFX_TRY(sor_window_loop(ctx, G, nx * ny, ny, P.TOL, P.maxiter, 2 * ny + nx - 2, HS_PLANE_C, batch, window, take,
                                    niter, error, 1, &L.sweep_hint, 0));
            OFX_TRY((op_skew<typename Pix<T>::v2, false>(ctx, L.Us, L.U, nx, ny, HS_PLANE_C_SKEW, G)));
        } else if (ctx->sor_exact && nx >= 3 && ny >= 3) {
            // one launch per time step (option sor_exact = 2): the reference implementation of the exact schedule
            const size_t ub = (size_t) nx * ny * sizeof(typename Pix<T>::v2);
            const dim3 gp(ofx_cdiv(ny + 3, 64), 1), bp(64);
            auto plane = [&](int t, int s_lo, int s_cnt) -> int {
                hipLaunchKernelGGL(k_hs_plane<T>, dim3(gp.x, s_cnt), bp, 0, ctx->stream, L.U, L.A, (const T *) L.Dif,
                                   ctx->d_err, t, s_lo, nx, ny, alpha2);
                OFX_LAUNCH_CHECK(ctx);
                return OFX_OK;
            };
            auto save = [&]() -> int {
                OFX_HIP(ctx, hipMemcpyAsync(L.Uck, L.U, ub, hipMemcpyDeviceToDevice, ctx->stream));
                return OFX_OK;
            };
            auto restore = [&]() -> int {
                OFX_HIP(ctx, hipMemcpyAsync(L.U, L.Uck, ub, hipMemcpyDeviceToDevice, ctx->stream));
                return OFX_OK;
            };
            OFX_TRY(sor_exact_loop(ctx, nx * ny, P.TOL, P.maxiter, 2 * ny + nx - 2, HS_PLANE_C, plane, save, restore,
                                   &niter[0], &error[0]));
        } else if (P.maxiter > 0 && error[0] > P.TOL) {
            LoopSpec LS;
            LS.max_iter = P.maxiter;
            LS.size = nx * ny;
            LS.thr = P.TOL;
            LS.crit = OFX_CRIT_SQRT_MEAN;
            LS.chunk = sor_pick_chunk(ctx, nx, ny, 4);
            LS.fixed = ctx->fixed_work != 0;
            LS.pairs = false;
            auto launch = [&](int k, int, double thr) -> int {
                for (int col = 0; col < 4; col++)
                    hipLaunchKernelGGL(k_hs_sor<T>, gc, b2d(), 0, ctx->stream, L.U, L.A, (const T *) L.Dif, ctx->d_err, k,
                                       nx, ny, col >> 1, col & 1, alpha2, thr);
                OFX_LAUNCH_CHECK(ctx);
                return OFX_OK;
            };
            OFX_TRY(ofx_run_loop(ctx, LS, launch, [](int) { return OFX_OK; }, &niter[0], &error[0], ctx->profile ? &ms : nullptr));
        }
        if (P.verbose && G == 1) fprintf(stderr, "Iterations %d (%g)\n", niter[0], error[0]);   // :233-235
        for (int g = 0; g < G; g++) {
            ofx_stats &S = stats[g];
            if (scale < OFX_MAX_SCALES) {
                if (w < OFX_MAX_SOLVES) { S.iters[scale][w] = niter[g]; S.error[scale][w] = error[g]; }
                S.iter_ms[scale] += ms;
                S.iter_launches[scale] += niter[g];
            }
            S.work_pix_iters += (double) niter[g] * nx * ny;
        }
    }
    return OFX_OK;
}

template <typename T>
static int upload_plane(ofx_ctx *ctx, const double *h, size_t n, T **out)
{
    double *stage;
    OFX_TRY(ofx_alloc(ctx, n, &stage));
    OFX_TRY(ofx_alloc(ctx, n, out));
    OFX_HIP(ctx, hipMemcpyAsync(stage, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return op_convert_in<T>(ctx, stage, *out, n);
}

template <typename T>
static int download_flow(ofx_ctx *ctx, const typename Pix<T>::v2 *U, double *u, double *v, size_t n)
{
    double *d1, *d2;
    OFX_TRY(ofx_alloc(ctx, n, &d1));
    OFX_TRY(ofx_alloc(ctx, n, &d2));
    OFX_TRY(op_deinterleave2<T>(ctx, U, d1, d2, n));
    OFX_HIP(ctx, hipMemcpyAsync(u, d1, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(v, d2, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

static void sor_stats_begin(ofx_stats *st, int nscales, int nsolves)
{
    memset(st, 0, sizeof(*st));
    st->nscales = nscales;
    st->nsolves = nsolves;
}
static void sor_stats_begin(ofx_ctx *ctx, int nscales, int nsolves) { sor_stats_begin(&ctx->stats, nscales, nsolves); }

template <typename T>
static int hs_single_scale_host(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                                const HsParams &P)
{
    const size_t n = (size_t) nx * ny;
    sor_stats_begin(ctx, 1, P.warps);
    ctx->stats.nx[0] = nx;
    ctx->stats.ny[0] = ny;
    HsLevel<T> L;
    OFX_TRY(hs_level_alloc<T>(ctx, L, nx, ny, 1, false));
    OFX_TRY(upload_plane<T>(ctx, I1, n, &L.I1));
    OFX_TRY(upload_plane<T>(ctx, I2, n, &L.I2));
    double *d1, *d2;
    OFX_TRY(ofx_alloc(ctx, n, &d1));
    OFX_TRY(ofx_alloc(ctx, n, &d2));
    OFX_HIP(ctx, hipMemcpyAsync(d1, u, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(d2, v, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_TRY(op_interleave2<T>(ctx, d1, d2, L.U, n));
    OFX_TRY(hs_single_scale_dev<T>(ctx, L, P, 0, &ctx->stats));
    return download_flow<T>(ctx, L.flow(0), u, v, n);
}

// src/horn_schunck_pyramidal.cpp:258-370 for G pairs in lockstep.  dI1[g] / dI2[g]: device images of storage type T.
// On success lv[0].flow(g) holds the flow of pair g.
template <typename T>
static int hs_pyramidal_dev(ofx_ctx *ctx, int G, const T *const *dI1, const T *const *dI2, int nx, int ny, const HsParams &P,
                            int nscales, double zfactor, std::vector<HsLevel<T>> &lv, ofx_stats *stats)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "hs: group of %d pairs", G);
    if (P.verbose && G == 1)
        fprintf(stderr, "Multiscale Horn-Schunck of a %dx%d pair\n\ta=%g ns=%d zf=%g nw=%d eps=%g mi=%d\n", nx, ny,
                P.alpha, nscales, zfactor, P.warps, P.TOL, P.maxiter);
    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nx, ny, nscales, zfactor, nxs, nys));
    for (int g = 0; g < G; g++) {
        sor_stats_begin(&stats[g], nscales, P.warps);
        for (int s = 0; s < nscales && s < OFX_MAX_SCALES; s++) { stats[g].nx[s] = nxs[s]; stats[g].ny[s] = nys[s]; }
    }
    lv.resize(nscales);
    for (int s = 0; s < nscales; s++) OFX_TRY(hs_level_alloc<T>(ctx, lv[s], nxs[s], nys[s], G, true));
    {                                                                                                  // :279-317
        T *tmpA, *tmpB;                                   // all 2 G images of the group per launch (op_build_pyramid_group)
        double *scr;
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nx * ny, &tmpA));
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nx * ny, &tmpB));
        OFX_TRY(ofx_alloc(ctx, (size_t) G * op_pyramid_scratch_doubles(), &scr));
        std::vector<T *> lA(nscales), lB(nscales);
        for (int s = 0; s < nscales; s++) { lA[s] = lv[s].I1; lB[s] = lv[s].I2; }
        OFX_TRY(op_build_pyramid_group<T>(ctx, G, (const void *const *) dI1, (const void *const *) dI2, nscales, zfactor,
                                          HS_PRESMOOTH_SIGMA, nxs.data(), nys.data(), lA.data(), lB.data(), tmpA, tmpB, scr));
    }
    HsLevel<T> &C = lv[nscales - 1];
    OFX_TRY(op_fill2<T>(ctx, C.U, C.n() * G));                                                          // :320-323
    for (int s = nscales - 1; s >= 0; s--) {                                                           // :326
        if (P.verbose && G == 1) fprintf(stderr, "Scale: %d %dx%d\n", s, lv[s].nx, lv[s].ny);
        OFX_TRY(hs_single_scale_dev<T>(ctx, lv[s], P, s, stats));
        if (!s) break;
        for (int g = 0; g < G; g++)
            OFX_TRY(op_zoom_in_flow<T>(ctx, lv[s].flow(g), lv[s - 1].U + g * lv[s - 1].n(), lv[s].nx, lv[s].ny,
                                       lv[s - 1].nx, lv[s - 1].ny, 1.0 / zfactor));                    // :345-352
    }
    return OFX_OK;
}

template <typename T>
static int hs_pyramidal_host(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                             const HsParams &P, int nscales, double zfactor)
{
    const size_t n = (size_t) nx * ny;
    T *dI1, *dI2;
    OFX_TRY(upload_plane<T>(ctx, I1, n, &dI1));
    OFX_TRY(upload_plane<T>(ctx, I2, n, &dI2));
    std::vector<HsLevel<T>> lv;
    const T *a = dI1, *b = dI2;
    OFX_TRY(hs_pyramidal_dev<T>(ctx, 1, &a, &b, nx, ny, P, nscales, zfactor, lv, &ctx->stats));
    return download_flow<T>(ctx, lv[0].flow(0), u, v, n);
}

template <typename T>
static int hs_group_devapi(ofx_ctx *ctx, int G, const void *const *dI1, const void *const *dI2, void *const *d_flo, int nx,
                           int ny, const HsParams &P, int nscales, double zfactor, ofx_stats *stats)
{
    std::vector<HsLevel<T>> lv;
    OFX_TRY(hs_pyramidal_dev<T>(ctx, G, (const T *const *) dI1, (const T *const *) dI2, nx, ny, P, nscales, zfactor, lv, stats));
    const size_t n = (size_t) nx * ny;
    for (int g = 0; g < G; g++) OFX_TRY(op_to_flo<T>(ctx, lv[0].flow(g), (float2 *) d_flo[g], n));
    return OFX_OK;
}

extern "C" int ofx_hs_single_scale(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nx,
                                   int ny, double alpha, int warps, double TOL, int maxiter, int verbose)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "hs: NULL pointer");
    if (nx < 2 || ny < 2 || warps < 1) return ofx_fail(ctx, OFX_ERR_ARG, "hs: bad size / warps");
    if (maxiter > OFX_HS_MAX_MAXITER) return ofx_fail(ctx, OFX_ERR_ARG, "hs: maxiter > %d", OFX_HS_MAX_MAXITER);
    const double t0 = ofx_now_ms();
    const HsParams P = {alpha, TOL, warps, maxiter, verbose};
    int s = ctx->precision == OFX_F64 ? hs_single_scale_host<double>(ctx, I1, I2, u, v, nx, ny, P)
                                      : hs_single_scale_host<float>(ctx, I1, I2, u, v, nx, ny, P);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

extern "C" int ofx_hs_pyramidal(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                                double alpha, int nscales, double zfactor, int warps, double TOL, int maxiter,
                                int verbose)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "hs: NULL pointer");
    if (warps < 1) return ofx_fail(ctx, OFX_ERR_ARG, "hs: warps=%d", warps);
    if (maxiter > OFX_HS_MAX_MAXITER) return ofx_fail(ctx, OFX_ERR_ARG, "hs: maxiter > %d", OFX_HS_MAX_MAXITER);
    const double t0 = ofx_now_ms();
    const HsParams P = {alpha, TOL, warps, maxiter, verbose};
    int s = ctx->precision == OFX_F64 ? hs_pyramidal_host<double>(ctx, I1, I2, u, v, nx, ny, P, nscales, zfactor)
                                      : hs_pyramidal_host<float>(ctx, I1, I2, u, v, nx, ny, P, nscales, zfactor);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

extern "C" int ofx_hs_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI1, const void *const *dI2,
                                void *const *d_flo, int nx, int ny, double alpha, int nscales, double zfactor, int warps,
                                double TOL, int maxiter, ofx_stats *stats_out)
{
    OFX_ENTER(ctx);
    if (!dI1 || !dI2 || !d_flo) return ofx_fail(ctx, OFX_ERR_ARG, "hs: NULL pointer");
    if (n_pairs < 1 || n_pairs > OFX_MAX_GROUP)
        return ofx_fail(ctx, OFX_ERR_ARG, "hs: a lockstep group holds 1..%d pairs (got %d)", OFX_MAX_GROUP, n_pairs);
    for (int g = 0; g < n_pairs; g++)
        if (!dI1[g] || !dI2[g] || !d_flo[g]) return ofx_fail(ctx, OFX_ERR_ARG, "hs: NULL pointer (pair %d)", g);
    if (warps < 1) return ofx_fail(ctx, OFX_ERR_ARG, "hs: warps=%d", warps);
    if (maxiter > OFX_HS_MAX_MAXITER) return ofx_fail(ctx, OFX_ERR_ARG, "hs: maxiter > %d", OFX_HS_MAX_MAXITER);
    const double t0 = ofx_now_ms();
    const HsParams P = {alpha, TOL, warps, maxiter, 0};
    std::vector<ofx_stats> local(stats_out ? 0 : n_pairs);
    ofx_stats *st = stats_out ? stats_out : local.data();
    int s = ctx->precision == OFX_F64 ? hs_group_devapi<double>(ctx, n_pairs, dI1, dI2, d_flo, nx, ny, P, nscales, zfactor, st)
                                      : hs_group_devapi<float>(ctx, n_pairs, dI1, dI2, d_flo, nx, ny, P, nscales, zfactor, st);
    const double ms = ofx_now_ms() - t0;
    for (int g = 0; g < n_pairs; g++) st[g].total_ms = ms;
    ctx->stats = st[0];
    return s;
}

// ============================================================================================
// Brox spatial
// ============================================================================================

// clamped centred differences / second derivatives at one pixel (src/operators.cpp:132-406, nz = 1)
template <typename T> OFX_DEV double cdx(const T *f, int i, int j, int nx, int ny)
{
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    return 0.5 * (ldw(f + (size_t) i * nx + jr) - ldw(f + (size_t) i * nx + jl));
}
template <typename T> OFX_DEV double cdy(const T *f, int i, int j, int nx, int ny)
{
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    return 0.5 * (ldw(f + (size_t) id * nx + j) - ldw(f + (size_t) iu * nx + j));
}
template <typename T> OFX_DEV double d2xx(const T *f, int i, int j, int nx, int ny)
{
    const size_t p = (size_t) i * nx + j;
    if (j == 0) return ldw(f + p) * -1.0 + ldw(f + p + 1);
    if (j == nx - 1) return ldw(f + p - 1) + ldw(f + p) * -1.0;
    return ldw(f + p - 1) + ldw(f + p) * -2.0 + ldw(f + p + 1);
}
template <typename T> OFX_DEV double d2yy(const T *f, int i, int j, int nx, int ny)
{
    const size_t p = (size_t) i * nx + j;
    if (i == 0) return ldw(f + p) * -1.0 + ldw(f + p + nx);
    if (i == ny - 1) return ldw(f + p - nx) + ldw(f + p) * -1.0;
    return ldw(f + p - nx) + ldw(f + p) * -2.0 + ldw(f + p + nx);
}
template <typename T> OFX_DEV double d2xy(const T *f, int i, int j, int nx, int ny)
{
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    return ldw(f + (size_t) iu * nx + jl) * 0.25 + ldw(f + (size_t) iu * nx + jr) * -0.25 +
           ldw(f + (size_t) id * nx + jl) * -0.25 + ldw(f + (size_t) id * nx + jr) * 0.25;
}

// src/brox_optic_flow_spatial.cpp:235-241: gradient of I1; I2 with its first and second derivatives
template <typename T>
__global__ void k_brox_prepare(const T *__restrict__ I1, const T *__restrict__ I2, typename Pix<T>::v2 *__restrict__ G1,
                               typename Pix<T>::v4 *__restrict__ PA, typename Pix<T>::v2 *__restrict__ PB, int nx, int ny)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t go = (size_t) blockIdx.z * nx * ny;             // pair of a lockstep group
    I1 += go; I2 += go; G1 += go; PA += go; PB += go;
    const size_t p = (size_t) i * nx + j;
    stn2(G1 + p, make_double2(cdx(I1, i, j, nx, ny), cdy(I1, i, j, nx, ny)));
    stn4(PA + p, make_double4(ldw(I2 + p), cdx(I2, i, j, nx, ny), cdy(I2, i, j, nx, ny), d2xx(I2, i, j, nx, ny)));
    stn2(PB + p, make_double2(d2xy(I2, i, j, nx, ny), d2yy(I2, i, j, nx, ny)));
}

// six bicubic warps with one set of taps, :246-251
template <typename T>
__global__ void k_brox_warp(const typename Pix<T>::v4 *__restrict__ PA, const typename Pix<T>::v2 *__restrict__ PB,
                            const typename Pix<T>::v2 *__restrict__ U, typename Pix<T>::v4 *__restrict__ WA,
                            typename Pix<T>::v2 *__restrict__ WB, int nx, int ny)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t go = (size_t) blockIdx.z * nx * ny;             // pair of a lockstep group
    PA += go; PB += go; U += go; WA += go; WB += go;
    const size_t p = (size_t) i * nx + j;
    const double2 u = ldw2(U + p);
    const BicubicTaps t = bicubic_taps(j + u.x, i + u.y, nx, ny);
    double4 wa = make_double4(0.0, 0.0, 0.0, 0.0);
    double2 wb = make_double2(0.0, 0.0);
    if (!t.out) {
        double c[6][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double4 a[4];
            double2 b[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a[r] = ldw4(PA + (size_t) t.row[r] * nx + t.col[k]);
                b[r] = ldw2(PB + (size_t) t.row[r] * nx + t.col[k]);
            }
            c[0][k] = cubic_cell(a[0].x, a[1].x, a[2].x, a[3].x, t.fy);
            c[1][k] = cubic_cell(a[0].y, a[1].y, a[2].y, a[3].y, t.fy);
            c[2][k] = cubic_cell(a[0].z, a[1].z, a[2].z, a[3].z, t.fy);
            c[3][k] = cubic_cell(a[0].w, a[1].w, a[2].w, a[3].w, t.fy);
            c[4][k] = cubic_cell(b[0].x, b[1].x, b[2].x, b[3].x, t.fy);
            c[5][k] = cubic_cell(b[0].y, b[1].y, b[2].y, b[3].y, t.fy);
        }
        wa.x = cubic_cell(c[0][0], c[0][1], c[0][2], c[0][3], t.fx);
        wa.y = cubic_cell(c[1][0], c[1][1], c[1][2], c[1][3], t.fx);
        wa.z = cubic_cell(c[2][0], c[2][1], c[2][2], c[2][3], t.fx);
        wa.w = cubic_cell(c[3][0], c[3][1], c[3][2], c[3][3], t.fx);
        wb.x = cubic_cell(c[4][0], c[4][1], c[4][2], c[4][3], t.fx);
        wb.y = cubic_cell(c[5][0], c[5][1], c[5][2], c[5][3], t.fx);
    }
    stn4(WA + p, wa);
    stn2(WB + p, wb);
}

// psi_smooth of the centred flow gradient, :254-258 + :99-122
// Expo != nullptr: robust_expo_psi_smooth (src/robust_expo_smoothness.cpp:26-44), the flow gradient weighed by expo
template <typename T>
__global__ void k_brox_psis(const typename Pix<T>::v2 *__restrict__ U, T *__restrict__ Psis, int nx, int ny,
                            const T *__restrict__ Expo = nullptr)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t go = (size_t) blockIdx.z * nx * ny;             // pair of a lockstep group
    U += go; Psis += go;
    if (Expo) Expo += go;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const double2 r = ldw2(U + (size_t) i * nx + jr), l = ldw2(U + (size_t) i * nx + jl);
    const double2 d = ldw2(U + (size_t) id * nx + j), t = ldw2(U + (size_t) iu * nx + j);
    const double ux = 0.5 * (r.x - l.x), uy = 0.5 * (d.x - t.x);
    const double vx = 0.5 * (r.y - l.y), vy = 0.5 * (d.y - t.y);
    if (Expo) {
        const double e = ldw(Expo + (size_t) i * nx + j);
        const double du = e * ux * ux + e * uy * uy;
        const double dv = e * vx * vx + e * vy * vy;
        const double normFlow = du + dv;
        stn(Psis + (size_t) i * nx + j, e / sqrt(normFlow + BROX_EPSILON * BROX_EPSILON));       // ROBUST_EXPO_EPSILON = 0.001 too
        return;
    }
    const double du = ux * ux + uy * uy;
    const double dv = vx * vx + vy * vy;
    const double d2 = du + dv;
    stn(Psis + (size_t) i * nx + j, 1. / sqrt(d2 + BROX_EPSILON * BROX_EPSILON));
}

// psi1..4 of src/brox_spatial_mask.cpp:16-93 at one pixel (0 across the image border)
struct Psi4 { double p1, p2, p3, p4; };
template <typename T, class Lay> OFX_DEV Psi4 brox_psi4_lay(const T *Psis, const Lay &lay, int i, int j, int nx, int ny)
{
    const double c = ldw(Psis + lay.idx(i, j));
    Psi4 r;
    r.p1 = (i < ny - 1) ? 0.5 * (ldw(Psis + lay.idx(i + 1, j)) + c) : 0.0;
    r.p2 = (i > 0) ? 0.5 * (ldw(Psis + lay.idx(i - 1, j)) + c) : 0.0;
    r.p3 = (j < nx - 1) ? 0.5 * (ldw(Psis + lay.idx(i, j + 1)) + c) : 0.0;
    r.p4 = (j > 0) ? 0.5 * (ldw(Psis + lay.idx(i, j - 1)) + c) : 0.0;
    return r;
}
template <typename T> OFX_DEV Psi4 brox_psi4(const T *Psis, int i, int j, int nx, int ny)
{
    return brox_psi4_lay(Psis, LayRow{nx}, i, j, nx, ny);
}

// div_u, div_v (src/brox_spatial_mask.cpp:100-171), div_d and du = dv = 0 (:261-274)
template <typename T>
__global__ void k_brox_div(const typename Pix<T>::v2 *__restrict__ U, const T *__restrict__ Psis,
                           typename Pix<T>::v2 *__restrict__ DV, T *__restrict__ Dd, typename Pix<T>::v2 *__restrict__ DU,
                           int nx, int ny, double alpha, int rx = 0)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t go = (size_t) blockIdx.z * nx * ny;             // pair of a lockstep group
    U += go; Psis += go; DV += go; Dd += go; DU += go;
    const size_t k = (size_t) i * nx + j;
    const Psi4 s = brox_psi4(Psis, i, j, nx, ny);
    const double2 c = ldw2(U + k);
    if (rx) {
        // robust_expo_divergence / div_d (src/robust_expo_generic_tensor.cpp:97-168, robust_expo_methods.cpp:266): the same
        // terms in the order right, left, down, up
        double du = 0.0, dv = 0.0;
        bool have = false;
        if (j < nx - 1) { const double2 q = ldw2(U + k + 1); du = s.p3 * (q.x - c.x); dv = s.p3 * (q.y - c.y); have = true; }
        if (j > 0) {
            const double2 q = ldw2(U + k - 1);
            const double a = s.p4 * (q.x - c.x), b = s.p4 * (q.y - c.y);
            du = have ? du + a : a; dv = have ? dv + b : b; have = true;
        }
        if (i < ny - 1) {
            const double2 q = ldw2(U + k + nx);
            const double a = s.p1 * (q.x - c.x), b = s.p1 * (q.y - c.y);
            du = have ? du + a : a; dv = have ? dv + b : b; have = true;
        }
        if (i > 0) {
            const double2 q = ldw2(U + k - nx);
            const double a = s.p2 * (q.x - c.x), b = s.p2 * (q.y - c.y);
            du = have ? du + a : a; dv = have ? dv + b : b; have = true;
        }
        stn2(DV + k, make_double2(du, dv));
        stn(Dd + k, alpha * (s.p3 + s.p4 + s.p1 + s.p2));
        stn2(DU + k, make_double2(0.0, 0.0));
        return;
    }
    // terms that exist, accumulated in the order down, up, right, left; missing terms are left out
    double du = 0.0, dv = 0.0;
    bool have = false;
    if (i < ny - 1) { const double2 q = ldw2(U + k + nx); du = s.p1 * (q.x - c.x); dv = s.p1 * (q.y - c.y); have = true; }
    if (i > 0) {
        const double2 q = ldw2(U + k - nx);
        const double a = s.p2 * (q.x - c.x), b = s.p2 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    if (j < nx - 1) {
        const double2 q = ldw2(U + k + 1);
        const double a = s.p3 * (q.x - c.x), b = s.p3 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    if (j > 0) {
        const double2 q = ldw2(U + k - 1);
        const double a = s.p4 * (q.x - c.x), b = s.p4 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    stn2(DV + k, make_double2(du, dv));
    stn(Dd + k, alpha * (s.p1 + s.p2 + s.p3 + s.p4));                  // :270
    stn2(DU + k, make_double2(0.0, 0.0));                               // :273
}

// psi_data, psi_gradient and the constant parts of the scheme, :279-309 (+ :33-92)
template <typename T>
__global__ void k_brox_coeff(const T *__restrict__ I1, const typename Pix<T>::v2 *__restrict__ G1,
                             const typename Pix<T>::v4 *__restrict__ WA, const typename Pix<T>::v2 *__restrict__ WB,
                             const typename Pix<T>::v2 *__restrict__ DU, const typename Pix<T>::v2 *__restrict__ DV,
                             const T *__restrict__ Dd, typename Pix<T>::v4 *__restrict__ CO, T *__restrict__ Dm, int n,
                             double alpha, double gamma, int rx = 0)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) n) return;
    const double i1 = ldw(I1 + i);
    const double2 g1 = ldw2(G1 + i);
    const double4 wa = ldw4(WA + i);
    const double2 wb = ldw2(WB + i);
    const double2 d = ldw2(DU + i);
    const double I2w = wa.x, I2wx = wa.y, I2wy = wa.z, I2wxx = wa.w, I2wxy = wb.x, I2wyy = wb.y;
    const double eps2 = BROX_EPSILON * BROX_EPSILON;
    if (rx) {
        // robust_expo_methods.cpp:48-60, :85-102, :279-322 for one channel: the motion terms are added before I1 is subtracted,
        // every per-channel sum is accumulated onto 0, and psi multiplies the finished sums
        const double dI = I2w + I2wx * d.x + I2wy * d.y - i1;
        const double psid = rnd_to<T>(1. / sqrt((0.0 + dI * dI) + eps2));
        const double dIx = I2wx + I2wxx * d.x + I2wxy * d.y - g1.x;
        const double dIy = I2wy + I2wxy * d.x + I2wyy * d.y - g1.y;
        const double psig = rnd_to<T>(1. / sqrt((0.0 + (dIx * dIx + dIy * dIy)) + eps2));
        const double dif = I2w - i1;
        double BNu = 0.0 + dif * I2wx, BNv = 0.0 + dif * I2wy, BDu = 0.0 + I2wx * I2wx, BDv = 0.0 + I2wy * I2wy;
        const double DI_Data = 0.0 + (I2wy * I2wx);
        const double dx = (I2wx - g1.x), dy = (I2wy - g1.y);
        double GNu = 0.0 + (dx * I2wxx + dy * I2wxy), GNv = 0.0 + (dx * I2wxy + dy * I2wyy);
        double GDu = 0.0 + (I2wxx * I2wxx + I2wxy * I2wxy), GDv = 0.0 + (I2wyy * I2wyy + I2wxy * I2wxy);
        const double DI_Gradient = 0.0 + (I2wxx + I2wyy) * I2wxy;
        const double g = gamma * psig;
        BNu = -psid * BNu; BNv = -psid * BNv; BDu = psid * BDu; BDv = psid * BDv;
        GNu = -g * GNu; GNv = -g * GNv; GDu = g * GDu; GDv = g * GDv;
        const double2 dvv = ldw2(DV + i);
        const double dd = ldw(Dd + i);
        stn4(CO + i, make_double4(BNu + GNu + alpha * dvv.x, BNv + GNv + alpha * dvv.y, BDu + GDu + dd, BDv + GDv + dd));
        stn(Dm + i, psid * DI_Data + g * DI_Gradient);
        return;
    }
    const double dI = I2w - i1 + I2wx * d.x + I2wy * d.y;                              // psi_data :51
    const double psid = rnd_to<T>(1. / sqrt(dI * dI + eps2));
    const double dIx = I2wx - g1.x + I2wxx * d.x + I2wxy * d.y;                        // psi_gradient :86-87
    const double dIy = I2wy - g1.y + I2wxy * d.x + I2wyy * d.y;
    const double dI2 = dIx * dIx + dIy * dIy;
    const double psig = rnd_to<T>(1. / sqrt(dI2 + eps2));
    const double p = psid;
    const double g = gamma * psig;
    const double dif = I2w - i1;
    const double BNu = -p * dif * I2wx;
    const double BNv = -p * dif * I2wy;
    const double BDu = p * I2wx * I2wx;
    const double BDv = p * I2wy * I2wy;
    const double dx = (I2wx - g1.x);
    const double dy = (I2wy - g1.y);
    const double GNu = -g * (dx * I2wxx + dy * I2wxy);
    const double GNv = -g * (dx * I2wxy + dy * I2wyy);
    const double GDu = g * (I2wxx * I2wxx + I2wxy * I2wxy);
    const double GDv = g * (I2wyy * I2wyy + I2wxy * I2wxy);
    const double DI = (I2wxx + I2wyy) * I2wxy;
    const double Duv = p * I2wy * I2wx + g * DI;
    const double2 dv = ldw2(DV + i);
    const double dd = ldw(Dd + i);
    stn4(CO + i, make_double4(BNu + GNu + alpha * dv.x, BNv + GNv + alpha * dv.y, BDu + GDu + dd, BDv + GDv + dd));
    stn(Dm + i, Duv);
}

// SOR update of one pixel, src/brox_optic_flow_spatial.cpp:129-172, in two halves (operands first, see HsOps);
// returns the squared update (:166)
struct BroxOps {
    Psi4    s;
    double2 c, dn, up, rt, lf;
    double4 co;
    double  D;
};
// psi1..4 through a sweep accessor (UGlobal: the arrays; BLds: the launch window in LDS); brox_psi4_lay's arithmetic
template <typename T, class Acc> OFX_DEV Psi4 brox_psi4_acc(const Acc &acc, const T *Psis, int i, int j, int nx, int ny)
{
    const double c = acc.psi(Psis, i, j);
    Psi4 r;
    r.p1 = (i < ny - 1) ? 0.5 * (acc.psi(Psis, i + 1, j) + c) : 0.0;
    r.p2 = (i > 0) ? 0.5 * (acc.psi(Psis, i - 1, j) + c) : 0.0;
    r.p3 = (j < nx - 1) ? 0.5 * (acc.psi(Psis, i, j + 1) + c) : 0.0;
    r.p4 = (j > 0) ? 0.5 * (acc.psi(Psis, i, j - 1) + c) : 0.0;
    return r;
}
template <typename T, class Acc>
OFX_DEV BroxOps brox_point_load(const Acc &acc, const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                                const T *__restrict__ Psis, int i, int j, int nx, int ny)
{
    BroxOps o;
    o.s = brox_psi4_acc<T>(acc, Psis, i, j, nx, ny);
    // a missing neighbour is addressed as the pixel itself (offset 0) with psi = 0, :332-388
    o.c = acc.get(i, j);
    o.dn = (i < ny - 1) ? acc.get(i + 1, j) : o.c;
    o.up = (i > 0) ? acc.get(i - 1, j) : o.c;
    o.rt = (j < nx - 1) ? acc.get(i, j + 1) : o.c;
    o.lf = (j > 0) ? acc.get(i, j - 1) : o.c;
    acc.coef4(CO, Dm, i, j, o.co, o.D);
    return o;
}
// RX: robust_expo_methods.cpp:138-148 -- the same taps summed right, left, down, up
template <typename T, class Acc, bool RX = false>
OFX_DEV double brox_point_finish(const Acc &acc, const BroxOps &o, int i, int j, double alpha)
{
    const Psi4 s = o.s;
    const double4 co = o.co;
    const double D = o.D;
    const double w = BROX_SOR_W;
    const double div_du = RX ? s.p3 * o.rt.x + s.p4 * o.lf.x + s.p1 * o.dn.x + s.p2 * o.up.x
                             : s.p1 * o.dn.x + s.p2 * o.up.x + s.p3 * o.rt.x + s.p4 * o.lf.x;      // :153-154
    const double div_dv = RX ? s.p3 * o.rt.y + s.p4 * o.lf.y + s.p1 * o.dn.y + s.p2 * o.up.y
                             : s.p1 * o.dn.y + s.p2 * o.up.y + s.p3 * o.rt.y + s.p4 * o.lf.y;      // :155-156
    const double duk = o.c.x, dvk = o.c.y;
    const double dun = rnd_to<T>((1. - w) * duk + w * (co.x - D * dvk + alpha * div_du) / co.z);   // :162
    const double dvn = rnd_to<T>((1. - w) * dvk + w * (co.y - D * dun + alpha * div_dv) / co.w);   // :163
    acc.put(i, j, make_double2(dun, dvn));
    return (dun - duk) * (dun - duk) + (dvn - dvk) * (dvn - dvk);                     // :166
}
template <typename T, class Acc, bool RX = false>
OFX_DEV double brox_point_acc(const Acc &acc, const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                              const T *__restrict__ Psis, int i, int j, int nx, int ny, double alpha)
{
    const BroxOps o = brox_point_load<T>(acc, CO, Dm, Psis, i, j, nx, ny);
    return brox_point_finish<T, Acc, RX>(acc, o, i, j, alpha);
}

template <typename T, bool COH = false, bool SNAP = false>
OFX_DEV double brox_point(typename Pix<T>::v2 *DU, const typename Pix<T>::v4 *__restrict__ CO,
                          const T *__restrict__ Dm, const T *__restrict__ Psis, int i, int j, int nx, int ny, double alpha,
                          typename Pix<T>::v2 *snap = nullptr)
{
    const UGlobal<T, COH, SNAP, LayRow> acc = {DU, snap, LayRow{nx}};
    return brox_point_acc<T>(acc, CO, Dm, Psis, i, j, nx, ny, alpha);
}
// the same on hyperplane-major arrays (windowed sweeps)
template <typename T, bool COH, bool SNAP>
OFX_DEV double brox_point_skew(typename Pix<T>::v2 *DU, const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                               const T *__restrict__ Psis, int i, int j, int nx, int ny, double alpha,
                               typename Pix<T>::v2 *snap)
{
    const UGlobal<T, COH, SNAP, LaySkew> acc = {DU, snap, LaySkew{ny, BROX_PLANE_C_SKEW}};
    return brox_point_acc<T>(acc, CO, Dm, Psis, i, j, nx, ny, alpha);
}

// one colour of one SOR sweep (fast, order-changing mode): every pixel with (i + j) % 2 == colour
// (blockIdx.z = pair of a lockstep group: its own error slots, err_stride doubles apart)
template <typename T>
__global__ __launch_bounds__(256) void k_brox_sor(typename Pix<T>::v2 *__restrict__ DU,
                                                  const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                                                  const T *__restrict__ Psis, double *__restrict__ err, int k, int nx,
                                                  int ny, int colour, double alpha, double tol, int err_stride)
{
    const size_t go = (size_t) blockIdx.z * nx * ny;
    DU += go; CO += go; Dm += go; Psis += go;
    err += (size_t) blockIdx.z * err_stride;
    const double prev = loop_fetch_prev(err, k);
    const int i = blockIdx.y * 4 + threadIdx.y;
    const int j = 2 * (blockIdx.x * 64 + threadIdx.x) + ((i + colour) & 1);
    const bool in = (j < nx) && (i < ny);
    const int gw = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.y;
    if (!loop_continues(prev, k, nx * ny, tol, OFX_CRIT_SQRT_MEAN)) return;
    double e = 0.0;
    if (in) e = brox_point<T>(DU, CO, Dm, Psis, i, j, nx, ny, alpha);
    loop_accumulate(err, k, e, gw);
}

// exact mode (see k_hs_plane): 5-point stencil -> pos = i + j in the interior, C = 2
// (visiting order brox_optic_flow_spatial.cpp:320-387)
#define BROX_PLANE_C 2
OFX_DEV bool brox_plane_item(int r, int q, int nx, int ny, int corner, int &i, int &j)
{
    if (r >= 1 && r <= ny - 2) { i = r; j = q - r; return j >= 1 && j <= nx - 2; }
    if (r == 0) { i = 0; j = q - 2; return j >= 1 && j <= nx - 2; }
    if (r == ny - 1) { i = ny - 1; j = q - (ny - 1); return j >= 1 && j <= nx - 2; }
    if (r == ny) { i = q - 2; j = 0; return i >= 1 && i <= ny - 2; }
    if (r == ny + 1) { i = q - (nx - 1); j = nx - 1; return i >= 1 && i <= ny - 2; }
    if (r == ny + 2) {
        if (corner == 0) { i = 0; j = 0; return q == 4; }
        if (corner == 1) { i = 0; j = nx - 1; return q == nx + 1; }
        if (corner == 2) { i = ny - 1; j = 0; return q == ny + 1; }
        i = ny - 1; j = nx - 1; return q == ny + nx - 2;
    }
    return false;
}

template <typename T>
__global__ __launch_bounds__(64) void k_brox_plane(typename Pix<T>::v2 *__restrict__ DU,
                                                   const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                                                   const T *__restrict__ Psis, double *__restrict__ err, int t, int s_lo,
                                                   int nx, int ny, double alpha)
{
    const int s = s_lo + blockIdx.y;
    const int q = t - BROX_PLANE_C * s;
    const int r = blockIdx.x * 64 + threadIdx.x;
    double e = 0.0;
    int i, j;
    if (r == ny + 2) {
        for (int corner = 0; corner < 4; corner++)
            if (brox_plane_item(r, q, nx, ny, corner, i, j)) e += brox_point<T>(DU, CO, Dm, Psis, i, j, nx, ny, alpha);
    } else if (brox_plane_item(r, q, nx, ny, 0, i, j)) {
        e = brox_point<T>(DU, CO, Dm, Psis, i, j, nx, ny, alpha);
    }
    loop_accumulate(err, s, e, blockIdx.x);
}

// windowed exact mode (see k_hs_window): K steps per launch, one workgroup per (SPW sweeps, row block)
template <typename T, int SPW, int MAXT, bool RX = false>
__global__ __launch_bounds__(MAXT) void k_brox_window(typename Pix<T>::v2 *DUg, typename Pix<T>::v2 *snap,
                                                      const typename Pix<T>::v4 *__restrict__ COg, const T *__restrict__ Dmg,
                                                      const T *__restrict__ Psisg, double *__restrict__ errg, SorWin w,
                                                      SorGrp grp, int s_cnt, int nx, int ny, double alpha)
{
    const int b = blockIdx.x, s0 = w.s_first + blockIdx.y * SPW, g = blockIdx.z;
    if (!((grp.runmask >> g) & 1u)) return;
    const int qmax = ny + nx - 2;
    int q_first[SPW];
    bool live[SPW], any = false;
#pragma unroll
    for (int u = 0; u < SPW; u++) {
        q_first[u] = w.tau0 - w.lag_s * (s0 + u) - w.lag_b * b;
        live[u] = (s0 + u < w.s_first + s_cnt) && !sor_unit_idle(w, b, q_first[u], w.K, 1, nx, ny);
        any = any || live[u];
    }
    if (!any) return;
    typename Pix<T>::v2 *DU = DUg + g * grp.npix;
    const typename Pix<T>::v4 *__restrict__ CO = COg + g * grp.npix;
    const T *__restrict__ Dm = Dmg + g * grp.npix;
    const T *__restrict__ Psis = Psisg + g * grp.npix;
    double *__restrict__ err = errg + (size_t) g * grp.err_stride;
    const int r = sor_window_item(w, b, threadIdx.x, ny);
    double e[SPW];
#pragma unroll
    for (int u = 0; u < SPW; u++) e[u] = 0.0;
    for (int k = 0; k < w.K; k++) {
        bool have[SPW];
        int pi[SPW], pj[SPW];
        BroxOps ops[SPW];
#pragma unroll
        for (int u = 0; u < SPW; u++) {
            const int q = q_first[u] + k;
            have[u] = live[u] && q >= 0 && q <= qmax && r >= 0 && r != ny + 2 && brox_plane_item(r, q, nx, ny, 0, pi[u], pj[u]) &&
                      (r < ny || sor_border_block(pi[u], ny, w.R) == b);
            if (SPW > 1 && have[u]) {
                const UGlobal<T, OFX_SOR_COH != 0, true, LaySkew> acc = {DU, snap + g * grp.snap_stride + (size_t) (s0 + u) * grp.npix,
                                                                         LaySkew{ny, BROX_PLANE_C_SKEW}};
                ops[u] = brox_point_load<T>(acc, CO, Dm, Psis, pi[u], pj[u], nx, ny);
            }
        }
#pragma unroll
        for (int u = 0; u < SPW; u++) {
            typename Pix<T>::v2 *mysnap = snap + g * grp.snap_stride + (size_t) (s0 + u) * grp.npix;
            if (have[u]) {
                const UGlobal<T, OFX_SOR_COH != 0, true, LaySkew> acc = {DU, mysnap, LaySkew{ny, BROX_PLANE_C_SKEW}};
                using AccT = UGlobal<T, OFX_SOR_COH != 0, true, LaySkew>;
                if (SPW > 1) e[u] += brox_point_finish<T, AccT, RX>(acc, ops[u], pi[u], pj[u], alpha);
                else e[u] += brox_point_acc<T, AccT, RX>(acc, CO, Dm, Psis, pi[u], pj[u], nx, ny, alpha);
            }
            const int q = q_first[u] + k;
            if (live[u] && r == ny + 2 && q >= 0 && q <= qmax) {
                int i, j;
                for (int corner = 0; corner < 4; corner++)
                    if (brox_plane_item(r, q, nx, ny, corner, i, j) && sor_border_block(i, ny, w.R) == b) {
                        using AccT = UGlobal<T, OFX_SOR_COH != 0, true, LaySkew>;
                        const AccT cacc = {DU, mysnap, LaySkew{ny, BROX_PLANE_C_SKEW}};
                        e[u] += brox_point_acc<T, AccT, RX>(cacc, CO, Dm, Psis, i, j, nx, ny, alpha);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < SPW; u++)
        if (live[u]) loop_accumulate(err, s0 + u, e[u], b * 4 + (threadIdx.x >> 6));
}

// The Brox sweep with its launch window in LDS (see k_hs_window_lds).  What the KW steps of (sweep, row block b) touch
// (tools/check_sor_schedule.py --cover): unknowns and psi_s on hyperplanes q0 - 4 .. q0 + KW of the skewed coordinate i + j,
// rows b R - 2 .. b R + R -- an LDS window each; (CO, Dm) of the KW pixels a thread updates -- LDS slots of its own.
template <typename T, int KW> struct BroxWinLds {
    static constexpr int NH = KW + 5;
    static constexpr int BACK = 4;
    static size_t bytes(int nr, int nt)
    {
        return (size_t) nt * KW * (sizeof(double4) + sizeof(double)) + (size_t) nr * NH * (sizeof(double2) + sizeof(double));
    }
};
template <typename T, bool SNAP> struct BLds {
    typename Pix<T>::v2 *U, *snap;
    double2 *win;                                // [NH][nr] (du, dv)
    const double *ps;                            // [NH][nr] psi_s
    int nr, row0, h0;
    LaySkew lay;
    double4 co;                                  // constant operands of the pixel being updated (set per step)
    double  D;
    OFX_DEV double2 get(int ii, int jj) const { return win[(ii + jj - h0) * nr + (ii - row0)]; }
    OFX_DEV double psi(const T *, int ii, int jj) const { return ps[(ii + jj - h0) * nr + (ii - row0)]; }
    OFX_DEV void coef4(const typename Pix<T>::v4 *, const T *, int, int, double4 &c, double &D_) const { c = co; D_ = D; }
    OFX_DEV void put(int i, int j, double2 v) const
    {
        win[(i + j - h0) * nr + (i - row0)] = v;
        const size_t p = lay.idx(i, j);
        stn2(U + p, v);
        if (SNAP) stn2(snap + p, v);
    }
};
OFX_DEV bool brox_window_pixel(const SorWin &w, int b, int r, int q, int qmax, int nx, int ny, int &pi, int &pj)
{
    return q >= 0 && q <= qmax && r >= 0 && r != ny + 2 && brox_plane_item(r, q, nx, ny, 0, pi, pj) &&
           (r < ny || sor_border_block(pi, ny, w.R) == b);
}
template <typename T, int KW, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_brox_window_lds(
    typename Pix<T>::v2 *DUg, typename Pix<T>::v2 *snap, const typename Pix<T>::v4 *__restrict__ COg, const T *__restrict__ Dmg,
    const T *__restrict__ Psisg, double *__restrict__ errg, SorWin w, SorGrp grp, int s_cnt, int nx, int ny, double alpha)
{
    using W = BroxWinLds<T, KW>;
    extern __shared__ double4 brox_lds[];
    const int b = blockIdx.x, s = w.s_first + blockIdx.y, g = blockIdx.z;
    if (!((grp.runmask >> g) & 1u)) return;
    const int qmax = ny + nx - 2;
    const int q0 = w.tau0 - w.lag_s * s - w.lag_b * b;
    if (sor_unit_idle(w, b, q0, KW, 1, nx, ny)) return;
    typename Pix<T>::v2 *DU = DUg + g * grp.npix;
    const typename Pix<T>::v4 *__restrict__ CO = COg + g * grp.npix;
    const T *__restrict__ Dm = Dmg + g * grp.npix;
    const T *__restrict__ Psis = Psisg + g * grp.npix;
    typename Pix<T>::v2 *mysnap = snap + g * grp.snap_stride + (size_t) s * grp.npix;
    double *__restrict__ err = errg + (size_t) g * grp.err_stride;
    const int nr = w.R + 3, row0 = b * w.R - 2, h0 = q0 - W::BACK, t = threadIdx.x, nt = blockDim.x;
    double4 *cco = brox_lds;                                     // [KW][nt]: 32-byte entries first (alignment)
    double2 *win = reinterpret_cast<double2 *>(cco + KW * nt);   // [NH][nr]
    double *ps = reinterpret_cast<double *>(win + W::NH * nr);   // [NH][nr]
    double *cdm = ps + W::NH * nr;                               // [KW][nt]
    const LaySkew lay = {ny, BROX_PLANE_C_SKEW};
    const int r = sor_window_item(w, b, t, ny);
    {
        const int i = row0 + t, ic = i < 0 ? 0 : (i > ny - 1 ? ny - 1 : i);
        const int hmax = (ny - 1) + nx - 1;
        double2 u_in[W::NH];
        double p_in[W::NH], d_in[KW];
        double4 c_in[KW];
#pragma unroll
        for (int k = 0; k < W::NH; k++) {
            const int h = h0 + k, hc = h < 0 ? 0 : (h > hmax ? hmax : h);
            u_in[k] = ldw2(DU + (size_t) hc * ny + ic);
            p_in[k] = ldw(Psis + (size_t) hc * ny + ic);
        }
#pragma unroll
        for (int k = 0; k < KW; k++) {
            int pi, pj;
            const bool have = brox_window_pixel(w, b, r, q0 + k, qmax, nx, ny, pi, pj);
            const size_t p = have ? lay.idx(pi, pj) : 0;
            c_in[k] = ldw4(CO + p);
            d_in[k] = ldw(Dm + p);
        }
        if (t < nr) {
#pragma unroll
            for (int k = 0; k < W::NH; k++) { win[k * nr + t] = u_in[k]; ps[k * nr + t] = p_in[k]; }
        }
#pragma unroll
        for (int k = 0; k < KW; k++) { cco[k * nt + t] = c_in[k]; cdm[k * nt + t] = d_in[k]; }
    }
    __syncthreads();
    BLds<T, true> acc = {DU, mysnap, win, ps, nr, row0, h0, lay, make_double4(0.0, 0.0, 0.0, 0.0), 0.0};
    double e = 0.0;
#pragma unroll 1
    for (int k = 0; k < KW; k++) {
        const int q = q0 + k;
        int pi, pj;
        if (brox_window_pixel(w, b, r, q, qmax, nx, ny, pi, pj)) {
            acc.co = cco[k * nt + t];
            acc.D = cdm[k * nt + t];
            e += brox_point_acc<T>(acc, CO, Dm, Psis, pi, pj, nx, ny, alpha);
        }
        if (q >= 0 && q <= qmax && r == ny + 2) {
            for (int corner = 0; corner < 4; corner++)
                if (brox_plane_item(r, q, nx, ny, corner, pi, pj) && sor_border_block(pi, ny, w.R) == b) {
                    const size_t p = lay.idx(pi, pj);
                    acc.co = ldw4(CO + p);
                    acc.D = ldw(Dm + p);
                    e += brox_point_acc<T>(acc, CO, Dm, Psis, pi, pj, nx, ny, alpha);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    loop_accumulate(err, s, e, b * 4 + (t >> 6));
}

// u += du, v += dv, :398-401
template <typename T>
__global__ void k_brox_add(typename Pix<T>::v2 *__restrict__ U, const typename Pix<T>::v2 *__restrict__ DU, int n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) n) return;
    const double2 u = ldw2(U + i), d = ldw2(DU + i);
    stn2(U + i, make_double2(u.x + d.x, u.y + d.y));
}

// One level of a lockstep group: every array holds G pairs back to back (pair g at element g * nx * ny).
template <typename T> struct BroxLevel {
    using v2 = typename Pix<T>::v2;
    using v4 = typename Pix<T>::v4;
    int nx, ny, G;
    T *I1, *I2, *Psis, *Dd, *Dm;
    v2 *G1, *PB, *WB, *U, *DV, *DU, *DUck;
    v4 *PA, *WA, *CO;
    // windowed exact mode, hyperplane-major (LaySkew), allocated on first use: (du, dv), the coefficients, psi_s and
    // snap_planes snapshot planes per pair
    v2 *DUs, *Snap;
    v4 *COs;
    T  *Dms, *Psiss;
    v2 *DUb = nullptr;    // tolerance mode (k_brox_wave): band-skewed copies of DU, CO, Dm, psi_s, allocated on first use
    v4 *COb = nullptr;
    T  *Dmb = nullptr, *Psb = nullptr;
    T  *Expo = nullptr;   // robust_expo: the smoothness weight of the level (allocated when needed)
    int snap_planes;
    int sweep_hint;
    size_t n() const { return (size_t) nx * ny; }
};

template <typename T> static int brox_level_alloc(ofx_ctx *ctx, BroxLevel<T> &L, int nx, int ny, int G)
{
    const size_t n = (size_t) nx * ny * G;
    L.nx = nx;
    L.ny = ny;
    L.G = G;
    OFX_TRY(ofx_alloc(ctx, n, &L.I1));
    OFX_TRY(ofx_alloc(ctx, n, &L.I2));
    OFX_TRY(ofx_alloc(ctx, n, &L.Psis));
    OFX_TRY(ofx_alloc(ctx, n, &L.Dd));
    OFX_TRY(ofx_alloc(ctx, n, &L.Dm));
    OFX_TRY(ofx_alloc(ctx, n, &L.G1));
    OFX_TRY(ofx_alloc(ctx, n, &L.PB));
    OFX_TRY(ofx_alloc(ctx, n, &L.WB));
    OFX_TRY(ofx_alloc(ctx, n, &L.U));
    OFX_TRY(ofx_alloc(ctx, n, &L.DV));
    OFX_TRY(ofx_alloc(ctx, n, &L.DU));
    OFX_TRY(ofx_alloc(ctx, n, &L.DUck));
    OFX_TRY(ofx_alloc(ctx, n, &L.PA));
    OFX_TRY(ofx_alloc(ctx, n, &L.WA));
    OFX_TRY(ofx_alloc(ctx, n, &L.CO));
    L.Snap = L.DUs = nullptr;
    L.COs = nullptr;
    L.Dms = L.Psiss = nullptr;
    L.snap_planes = 0;
    L.sweep_hint = 0;
    return OFX_OK;
}

struct BroxParams {
    double alpha, gamma, TOL;
    int inner_iter, outer_iter, verbose;
    // robust_expo_methods: Brox's scheme with the image-driven smoothness weight `expo` and that source's own
    // summation orders (the kernels' rx / RX variants); 0 = Brox
    int robust = 0, method = 1;
    double lambda = 0.0;
};

// src/brox_optic_flow_spatial.cpp:179-444 on device data for the G pairs of a lockstep group; stats[g] = record of pair g
// robust_expo_exponential_calculation (src/robust_expo_smoothness.cpp:128-187, one channel): expo = exp(-lambda |grad I1|)
// (+ 0.001 for method 2), method 3 with a per-pixel lambda bounded by the value at the 0.94 quantile of the sorted gradient
// magnitudes.  Evaluated on the HOST once per level: the reference calls libm's exp and log, whose last bit the device's
// math library does not promise to share, and method 3 sorts the level (std::sort there, here).  The gradient comes from the
// level's (I1x, I1y) pairs the prepare kernel has just written.
template <typename T> static int rexpo_level_expo(ofx_ctx *ctx, BroxLevel<T> &L, const BroxParams &P)
{
    const size_t n = L.n();
    if (!L.Expo) OFX_TRY(ofx_alloc(ctx, n, &L.Expo));
    std::vector<typename Pix<T>::v2> g1(n);
    std::vector<T> expo(n);
    OFX_HIP(ctx, hipMemcpyAsync(g1.data(), L.G1, n * sizeof(typename Pix<T>::v2), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> mg(n);
    for (size_t i = 0; i < n; i++) {
        const double ix = (double) g1[i].x, iy = (double) g1[i].y;
        mg[i] = sqrt(ix * ix + iy * iy);
    }
    if (P.method == 1 || P.method == 2) {
        const double beta = P.method == 2 ? 0.001 : 0.0;
        for (size_t i = 0; i < n; i++) expo[i] = (T) (exp(-P.lambda * mg[i]) + beta);
    } else {
        std::vector<double> ord(mg);
        std::sort(ord.begin(), ord.end());
        const double c = -log(0.05) + log(P.alpha);
        long pos_ref = (long) (int) (0.94 * (double) (int) n);
        double lambda_omega;
        while (pos_ref < (long) n && c / 2 > ord[pos_ref - 1]) pos_ref++;
        if (pos_ref == (long) n) lambda_omega = 0;
        else lambda_omega = c / ord[pos_ref - 1];
        for (size_t i = 0; i < n; i++) {
            const double lp = (-log(0.05) + log(P.alpha)) / mg[i];
            double lambda_pi = lambda_omega;
            if (lambda_omega > lp) lambda_pi = lp;
            expo[i] = (T) exp(-lambda_pi * mg[i]);
        }
    }
    OFX_HIP(ctx, hipMemcpyAsync(L.Expo, expo.data(), n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));               // `expo` is a local
    return OFX_OK;
}

template <typename T>
static int brox_single_scale_dev(ofx_ctx *ctx, BroxLevel<T> &L, const BroxParams &P, int scale, ofx_stats *stats)
{
    const int nx = L.nx, ny = L.ny, n = nx * ny, G = L.G;
    const size_t npix = L.n();
    const dim3 g(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G), b = b2d();
    const dim3 g1((unsigned) ((npix * G + 255) / 256)), b1(256);
    const dim3 gc(ofx_cdiv(ofx_cdiv(nx, 2) + 1, 64), ofx_cdiv(ny, 4));
    if ((long long) npix * G >= (1LL << 31)) return ofx_fail(ctx, OFX_ERR_ARG, "brox: group larger than 2^31 pixels");
    const bool windowed = ctx->sor_exact == 1 && nx >= 3 && ny >= 3;
    // tolerance mode (sor_exact = 0, ofx_sor_tile.hip): the finest levels sweep a checkerboard of tiles in the reference's order
    // inside a tile, the coarser ones red-black; both serve lockstep groups
    const bool tol_mode = ctx->sor_exact == 0 && ctx->sor_fuse >= 0 && !P.robust;
    const bool wave = tol_mode && scale < ctx->sor_wave_levels;
    if (G > 1 && !windowed && !tol_mode)
        return ofx_fail(ctx, OFX_ERR_ARG, "brox: lockstep groups need sor_exact = 1 (levels of at least 3x3, this one %dx%d) or the tile sweeps of sor_exact = 0", nx, ny);
    int solve = 0;
    const int rx = P.robust;
    if (rx && (!windowed || G != 1))
        return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: needs sor_exact = 1 and levels of at least 3x3 (%dx%d)", nx, ny);
    hipLaunchKernelGGL(k_brox_prepare<T>, g, b, 0, ctx->stream, (const T *) L.I1, (const T *) L.I2, L.G1, L.PA, L.PB, nx, ny);
    OFX_LAUNCH_CHECK(ctx);
    if (rx) OFX_TRY(rexpo_level_expo<T>(ctx, L, P));                                              // robust_expo_methods.cpp:231
    for (int no = 0; no < P.outer_iter; no++) {                                                   // :244
        hipLaunchKernelGGL(k_brox_warp<T>, g, b, 0, ctx->stream, L.PA, L.PB, L.U, L.WA, L.WB, nx, ny);
        hipLaunchKernelGGL(k_brox_psis<T>, g, b, 0, ctx->stream, L.U, L.Psis, nx, ny, (const T *) (rx ? L.Expo : nullptr));
        hipLaunchKernelGGL(k_brox_div<T>, g, b, 0, ctx->stream, L.U, (const T *) L.Psis, L.DV, L.Dd, L.DU, nx, ny, P.alpha, rx);
        OFX_LAUNCH_CHECK(ctx);
        for (int ni = 0; ni < P.inner_iter; ni++) {                                               // :277
            hipLaunchKernelGGL(k_brox_coeff<T>, g1, b1, 0, ctx->stream, (const T *) L.I1, L.G1, L.WA, L.WB, L.DU, L.DV,
                               (const T *) L.Dd, L.CO, L.Dm, (int) (npix * G), P.alpha, P.gamma, rx);
            OFX_LAUNCH_CHECK(ctx);
            int nsor[OFX_MAX_GROUP] = {0};
            double error[OFX_MAX_GROUP];
            for (int q = 0; q < G; q++) error[q] = 1000;                                          // :312
            float ms = 0.f;
            if (wave) {
                // band-skewed copies of DU, CO, Dm, psi_s (k_brox_wave)
                const size_t ps = ofx_band_plane_elems(nx, ny);
                if (!L.DUb) {
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.DUb));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.COb));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.Dmb));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.Psb));
                }
                OFX_TRY((ofx_band_copy<typename Pix<T>::v2, true>(ctx, L.DU, L.DUb, nx, ny, G)));
                OFX_TRY((ofx_band_copy<typename Pix<T>::v4, true>(ctx, L.CO, L.COb, nx, ny, G)));
                OFX_TRY((ofx_band_copy<T, true>(ctx, L.Dm, L.Dmb, nx, ny, G)));
                OFX_TRY((ofx_band_copy<T, true>(ctx, L.Psis, L.Psb, nx, ny, G)));
                OFX_TRY(ofx_brox_wave_solve<T>(ctx, G, L.DUb, L.COb, (const T *) L.Dmb, (const T *) L.Psb, nx, ny, P.alpha, P.TOL,
                                               OFX_BROX_MAX_ITERATIONS, nsor, error, ctx->profile ? &ms : nullptr));
                OFX_TRY((ofx_band_copy<typename Pix<T>::v2, false>(ctx, L.DUb, L.DU, nx, ny, G)));
            } else if (windowed) {
                // the sweeps run on hyperplane-major copies of DU, CO, Dm, psi_s
                const size_t ps = skew_plane_elems(nx, ny, BROX_PLANE_C_SKEW);
                const size_t ub = ps * sizeof(typename Pix<T>::v2);
                const int batch = sor_pick_batch(ctx, ps, sizeof(typename Pix<T>::v2), OFX_BROX_MAX_ITERATIONS, G);
                if (!L.DUs) {
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.DUs));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.COs));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.Dms));
                    OFX_TRY(ofx_alloc(ctx, ps * G, &L.Psiss));
                }
                if (L.snap_planes < batch) {
                    OFX_TRY(ofx_alloc(ctx, ps * batch * G, &L.Snap));
                    L.snap_planes = batch;
                }
                OFX_TRY((op_skew<typename Pix<T>::v2, true>(ctx, L.DU, L.DUs, nx, ny, BROX_PLANE_C_SKEW, G)));
                OFX_TRY((op_skew<typename Pix<T>::v4, true>(ctx, L.CO, L.COs, nx, ny, BROX_PLANE_C_SKEW, G)));
                OFX_TRY((op_skew<T, true>(ctx, L.Dm, L.Dms, nx, ny, BROX_PLANE_C_SKEW, G)));
                OFX_TRY((op_skew<T, true>(ctx, L.Psis, L.Psiss, nx, ny, BROX_PLANE_C_SKEW, G)));
                const size_t snap_stride = ps * L.snap_planes;
                auto window = [&](const SorWin &w, int blocks, int sweeps, unsigned runmask, int err_stride) -> int {
                    const SorGrp grp = {runmask, err_stride, ps, snap_stride};
                    if (rx) {                                    // robust_expo: the global window kernel with its summation order
                        hipLaunchKernelGGL((k_brox_window<T, 1, 1024, true>), dim3(blocks, sweeps, G), dim3(sor_window_threads(w.R + 3)),
                                           0, ctx->stream, L.DUs, L.Snap, L.COs, (const T *) L.Dms, (const T *) L.Psiss, ctx->d_err, w,
                                           grp, sweeps, nx, ny, P.alpha);
                        OFX_LAUNCH_CHECK(ctx);
                        return OFX_OK;
                    }
                    const int spw = sor_pick_spw(ctx, G);
                    const size_t lds_need = (size_t) sor_window_threads(w.R + 3) * w.K * (sizeof(double4) + sizeof(double)) +
                                            (size_t) (w.R + 3) * (w.K + 5) * (sizeof(double2) + sizeof(double));
                    if (sor_use_lds(ctx, G, true) && (w.K == 4 || w.K == 8 || w.K == 16) && w.R + 3 <= 256 && lds_need <= 160 * 1024) {
                        const dim3 grid(blocks, sweeps, G), blk(sor_window_threads(w.R + 3));
#define OFX_BROX_WINL(K_)                                                                                                      \
    do {                                                                                                                       \
        const size_t lds = BroxWinLds<T, K_>::bytes(w.R + 3, (int) blk.x);                                                     \
        if (lds > 160 * 1024) return ofx_fail(ctx, OFX_ERR_ARG, "brox: window of %d steps x %d rows does not fit the LDS", K_, w.R); \
        if (lds > 64 * 1024)                                                                                                   \
            OFX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_brox_window_lds<T, K_, 256>),                    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                          \
        hipLaunchKernelGGL((k_brox_window_lds<T, K_, 256>), grid, blk, lds, ctx->stream, L.DUs, L.Snap, L.COs, (const T *) L.Dms, \
                           (const T *) L.Psiss, ctx->d_err, w, grp, sweeps, nx, ny, P.alpha);                                  \
    } while (0)
                        if (w.K == 4) OFX_BROX_WINL(4);
                        else if (w.K == 8) OFX_BROX_WINL(8);
                        else OFX_BROX_WINL(16);
#undef OFX_BROX_WINL
                        OFX_LAUNCH_CHECK(ctx);
                        return OFX_OK;
                    }
                    const dim3 grid(blocks, ofx_cdiv(sweeps, spw), G), blk(sor_window_threads(w.R + 3));
#define OFX_BROX_WIN(SPW_, MAXT_)                                                                                        \
    hipLaunchKernelGGL((k_brox_window<T, SPW_, MAXT_>), grid, blk, 0, ctx->stream, L.DUs, L.Snap, L.COs, (const T *) L.Dms, \
                       (const T *) L.Psiss, ctx->d_err, w, grp, sweeps, nx, ny, P.alpha)
                    if (blk.x <= 128 && spw > 1) {
                        if (spw == 4) OFX_BROX_WIN(4, 128);
                        else OFX_BROX_WIN(2, 128);
                    } else {
                        if (spw == 4) OFX_BROX_WIN(4, 1024);
                        else if (spw == 2) OFX_BROX_WIN(2, 1024);
                        else OFX_BROX_WIN(1, 1024);
                    }
#undef OFX_BROX_WIN
                    OFX_LAUNCH_CHECK(ctx);
                    return OFX_OK;
                };
                auto take = [&](int q, int k) -> int {
                    OFX_HIP(ctx, hipMemcpyAsync(L.DUs + q * ps, L.Snap + q * snap_stride + (size_t) (k - 1) * ps, ub,
                                                hipMemcpyDeviceToDevice, ctx->stream));
                    return OFX_OK;
                };
                OFX_TRY(sor_window_loop(ctx, G, n, ny, P.TOL, OFX_BROX_MAX_ITERATIONS, ny + nx - 2, BROX_PLANE_C, batch, window,
                                        take, nsor, error, 1, &L.sweep_hint, 0));
                OFX_TRY((op_skew<typename Pix<T>::v2, false>(ctx, L.DUs, L.DU, nx, ny, BROX_PLANE_C_SKEW, G)));
            } else if (ctx->sor_exact && nx >= 3 && ny >= 3) {
                const size_t ub = (size_t) n * sizeof(typename Pix<T>::v2);
                const unsigned gpx = ofx_cdiv(ny + 3, 64);
                auto plane = [&](int t, int s_lo, int s_cnt) -> int {
                    hipLaunchKernelGGL(k_brox_plane<T>, dim3(gpx, s_cnt), dim3(64), 0, ctx->stream, L.DU, L.CO,
                                       (const T *) L.Dm, (const T *) L.Psis, ctx->d_err, t, s_lo, nx, ny, P.alpha);
                    OFX_LAUNCH_CHECK(ctx);
                    return OFX_OK;
                };
                auto save = [&]() -> int {
                    OFX_HIP(ctx, hipMemcpyAsync(L.DUck, L.DU, ub, hipMemcpyDeviceToDevice, ctx->stream));
                    return OFX_OK;
                };
                auto restore = [&]() -> int {
                    OFX_HIP(ctx, hipMemcpyAsync(L.DU, L.DUck, ub, hipMemcpyDeviceToDevice, ctx->stream));
                    return OFX_OK;
                };
                OFX_TRY(sor_exact_loop(ctx, n, P.TOL, OFX_BROX_MAX_ITERATIONS, ny + nx - 2, BROX_PLANE_C, plane, save,
                                       restore, &nsor[0], &error[0]));
            } else if (tol_mode && ctx->sor_fuse != 9) {
                // the coarser levels of the tolerance mode: red-black, K sweeps per launch on LDS tiles (k_brox_tile; sor_fuse = 9:
                // the two launches per sweep of k_brox_sor below, for A/B)
                OFX_TRY(ofx_brox_tile_solve<T>(ctx, G, L.DU, L.DUck, L.CO, (const T *) L.Dm, (const T *) L.Psis, nx, ny, P.alpha, P.TOL,
                                               OFX_BROX_MAX_ITERATIONS, ctx->sor_fuse, nsor, error, ctx->profile ? &ms : nullptr));
            } else if (error[0] > P.TOL) {
                LoopSpec LS;
                LS.max_iter = OFX_BROX_MAX_ITERATIONS;
                LS.size = n;
                LS.thr = P.TOL;
                LS.crit = OFX_CRIT_SQRT_MEAN;
                LS.chunk = sor_pick_chunk(ctx, nx, ny, 2);
                LS.fixed = ctx->fixed_work != 0;
                LS.pairs = false;
                const int err_stride = (LS.max_iter + 1) * OFX_NSHARD;
                const dim3 gcg(gc.x, gc.y, G);
                auto launch = [&](int k, int, double thr) -> int {
                    for (int col = 0; col < 2; col++)
                        hipLaunchKernelGGL(k_brox_sor<T>, gcg, b, 0, ctx->stream, L.DU, L.CO, (const T *) L.Dm,
                                           (const T *) L.Psis, ctx->d_err, k, nx, ny, col, P.alpha, thr, err_stride);
                    OFX_LAUNCH_CHECK(ctx);
                    return OFX_OK;
                };
                OFX_TRY(ofx_run_loop_group(ctx, LS, G, launch, [](const int *) { return OFX_OK; }, nsor, error, ctx->profile ? &ms : nullptr));
            }
            if (P.verbose && G == 1) {                                                            // :392-394 / robust :414-416
                if (rx) printf("Iterations: %d Error: %g\n", nsor[0], error[0]);
                else printf("Iterations: %d\n", nsor[0]);
                fflush(stdout);
            }
            for (int q = 0; q < G; q++) {
                ofx_stats &S = stats[q];
                if (scale < OFX_MAX_SCALES) {
                    if (solve < OFX_MAX_SOLVES) { S.iters[scale][solve] = nsor[q]; S.error[scale][solve] = error[q]; }
                    S.iter_ms[scale] += ms;
                    S.iter_launches[scale] += nsor[q];
                }
                S.work_pix_iters += (double) nsor[q] * n;
            }
            solve++;
        }
        hipLaunchKernelGGL(k_brox_add<T>, g1, b1, 0, ctx->stream, L.U, L.DU, (int) (npix * G));   // :398-401
        OFX_LAUNCH_CHECK(ctx);
    }
    return OFX_OK;
}

// src/brox_optic_flow_spatial.cpp:451-549 for G pairs in lockstep; on success lv[0].U holds the flows
template <typename T>
static int brox_spatial_dev(ofx_ctx *ctx, int G, const T *const *dI1, const T *const *dI2, int nx, int ny,
                            const BroxParams &P, int nscales, double nu, std::vector<BroxLevel<T>> &lv, ofx_stats *stats)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "brox: group of %d pairs", G);
    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nx, ny, nscales, nu, nxs, nys));
    for (int g = 0; g < G; g++) {
        sor_stats_begin(&stats[g], nscales, P.inner_iter * P.outer_iter);
        for (int s = 0; s < nscales && s < OFX_MAX_SCALES; s++) { stats[g].nx[s] = nxs[s]; stats[g].ny[s] = nys[s]; }
    }
    lv.resize(nscales);
    for (int s = 0; s < nscales; s++) OFX_TRY(brox_level_alloc<T>(ctx, lv[s], nxs[s], nys[s], G));
    {                                                                                              // :467-504
        T *tmpA, *tmpB;                                   // all 2 G images of the group per launch (op_build_pyramid_group)
        double *scr;
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nx * ny, &tmpA));
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nx * ny, &tmpB));
        OFX_TRY(ofx_alloc(ctx, (size_t) G * op_pyramid_scratch_doubles(), &scr));
        std::vector<T *> lA(nscales), lB(nscales);
        for (int s = 0; s < nscales; s++) { lA[s] = lv[s].I1; lB[s] = lv[s].I2; }
        if (P.robust) {
            // robust_expo_methods.cpp:494-525 for one channel: image_normalization_2_color == image_normalization_2, then
            // gaussian(I, nx, ny, nzz, GAUSSIAN_SIGMA) -- i.e. sigma = the number of channels = 1 and boundary condition
            // (int) 0.8 = BOUNDARY_CONDITION_DIRICHLET --, then zoom_out_color == zoom_out
            OFX_TRY(op_normalize2<T>(ctx, dI1[0], dI2[0], lA[0], lB[0], nx * ny, scr));
            OFX_TRY(op_gaussian<T>(ctx, lA[0], tmpA, nx, ny, 1.0, 1));
            OFX_TRY(op_gaussian<T>(ctx, lB[0], tmpA, nx, ny, 1.0, 1));
            for (int s = 1; s < nscales; s++) {
                OFX_TRY(op_zoom_out<T>(ctx, lA[s - 1], lA[s], tmpA, tmpB, nxs[s - 1], nys[s - 1], nu));
                OFX_TRY(op_zoom_out<T>(ctx, lB[s - 1], lB[s], tmpA, tmpB, nxs[s - 1], nys[s - 1], nu));
            }
        } else
        OFX_TRY(op_build_pyramid_group<T>(ctx, G, (const void *const *) dI1, (const void *const *) dI2, nscales, nu,
                                          BROX_SIGMA, nxs.data(), nys.data(), lA.data(), lB.data(), tmpA, tmpB, scr));
    }
    BroxLevel<T> &C = lv[nscales - 1];
    OFX_TRY(op_fill2<T>(ctx, C.U, C.n() * G));                                                     // :507-509
    for (int s = nscales - 1; s >= 0; s--) {                                                      // :516
        if (P.verbose && G == 1) { printf("Scale: %d\n", s); fflush(stdout); }
        OFX_TRY(brox_single_scale_dev<T>(ctx, lv[s], P, s, stats));
        if (s)
            for (int g = 0; g < G; g++)
                OFX_TRY(op_zoom_in_flow<T>(ctx, lv[s].U + g * lv[s].n(), lv[s - 1].U + g * lv[s - 1].n(), lv[s].nx, lv[s].ny,
                                           lv[s - 1].nx, lv[s - 1].ny, 1.0 / nu));                // :529-535
    }
    return OFX_OK;
}

template <typename T>
static int brox_spatial_host(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                             const BroxParams &P, int nscales, double nu)
{
    const size_t n = (size_t) nx * ny;
    T *dI1, *dI2;
    OFX_TRY(upload_plane<T>(ctx, I1, n, &dI1));
    OFX_TRY(upload_plane<T>(ctx, I2, n, &dI2));
    std::vector<BroxLevel<T>> lv;
    const T *a = dI1, *b = dI2;
    OFX_TRY(brox_spatial_dev<T>(ctx, 1, &a, &b, nx, ny, P, nscales, nu, lv, &ctx->stats));
    return download_flow<T>(ctx, lv[0].U, u, v, n);
}

template <typename T>
static int brox_group_devapi(ofx_ctx *ctx, int G, const void *const *dI1, const void *const *dI2, void *const *d_flo, int nx,
                             int ny, const BroxParams &P, int nscales, double nu, ofx_stats *stats)
{
    std::vector<BroxLevel<T>> lv;
    OFX_TRY(brox_spatial_dev<T>(ctx, G, (const T *const *) dI1, (const T *const *) dI2, nx, ny, P, nscales, nu, lv, stats));
    const size_t n = (size_t) nx * ny;
    for (int g = 0; g < G; g++) OFX_TRY(op_to_flo<T>(ctx, lv[0].U + g * n, (float2 *) d_flo[g], n));
    return OFX_OK;
}

extern "C" int ofx_brox_spatial(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nxx,
                                int nyy, double alpha, double gamma, int nscales, double nu, double TOL,
                                int inner_iter, int outer_iter, int verbose)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "brox: NULL pointer");
    if (inner_iter < 0 || outer_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "brox: negative iteration count");
    const double t0 = ofx_now_ms();
    const BroxParams P = {alpha, gamma, TOL, inner_iter, outer_iter, verbose};
    int s = ctx->precision == OFX_F64 ? brox_spatial_host<double>(ctx, I1, I2, u, v, nxx, nyy, P, nscales, nu)
                                      : brox_spatial_host<float>(ctx, I1, I2, u, v, nxx, nyy, P, nscales, nu);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

// robust_expo_methods (src/robust_expo_methods.h:21-38), one channel
extern "C" int ofx_robust_expo(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nxx, int nyy, int nzz,
                               int method_type, double alpha, double gamma, double lambda, int nscales, double nu, double TOL,
                               int inner_iter, int outer_iter, int verbose)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: NULL pointer");
    if (nzz != 1)
        return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: nzz=%d (one channel only: for colour the reference's pyramid reads beyond its "
                                          "scratch copy, zoom.cpp:96-118)", nzz);
    if (method_type < 1 || method_type > 3) return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: method_type=%d (1, 2 or 3)", method_type);
    if (inner_iter < 0 || outer_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: negative iteration count");
    if (ctx->sor_exact != 1) return ofx_fail(ctx, OFX_ERR_ARG, "robust_expo: needs the default option sor_exact = 1");
    const double t0 = ofx_now_ms();
    BroxParams P = {(double) (int) (alpha * nzz), gamma, TOL, inner_iter, outer_iter, verbose};     // :529: alpha * nzz as an int
    P.robust = 1;
    P.method = method_type;
    P.lambda = lambda;
    int s = ctx->precision == OFX_F64 ? brox_spatial_host<double>(ctx, I1, I2, u, v, nxx, nyy, P, nscales, nu)
                                      : brox_spatial_host<float>(ctx, I1, I2, u, v, nxx, nyy, P, nscales, nu);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

extern "C" int ofx_brox_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI1, const void *const *dI2,
                                  void *const *d_flo, int nxx, int nyy, double alpha, double gamma, int nscales, double nu,
                                  double TOL, int inner_iter, int outer_iter, ofx_stats *stats_out)
{
    OFX_ENTER(ctx);
    if (!dI1 || !dI2 || !d_flo) return ofx_fail(ctx, OFX_ERR_ARG, "brox: NULL pointer");
    if (n_pairs < 1 || n_pairs > OFX_MAX_GROUP)
        return ofx_fail(ctx, OFX_ERR_ARG, "brox: a lockstep group holds 1..%d pairs (got %d)", OFX_MAX_GROUP, n_pairs);
    for (int g = 0; g < n_pairs; g++)
        if (!dI1[g] || !dI2[g] || !d_flo[g]) return ofx_fail(ctx, OFX_ERR_ARG, "brox: NULL pointer (pair %d)", g);
    if (inner_iter < 0 || outer_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "brox: negative iteration count");
    const double t0 = ofx_now_ms();
    const BroxParams P = {alpha, gamma, TOL, inner_iter, outer_iter, 0};
    std::vector<ofx_stats> local(stats_out ? 0 : n_pairs);
    ofx_stats *st = stats_out ? stats_out : local.data();
    int s = ctx->precision == OFX_F64 ? brox_group_devapi<double>(ctx, n_pairs, dI1, dI2, d_flo, nxx, nyy, P, nscales, nu, st)
                                      : brox_group_devapi<float>(ctx, n_pairs, dI1, dI2, d_flo, nxx, nyy, P, nscales, nu, st);
    const double ms = ofx_now_ms() - t0;
    for (int g = 0; g < n_pairs; g++) st[g].total_ms = ms;
    ctx->stats = st[0];
    return s;
}

// ---- batches of pairs for the SOR solvers: lockstep groups, one worker thread per context (as ofx_tvl1_batch_dev) ----
template <class GroupFn>
static int sor_batch_run(ofx_ctx *const *ctxs, int n_ctx, int n_pairs, int G, double *work_pix_iters, GroupFn group)
{
    const int n_groups = (n_pairs + G - 1) / G;
    std::atomic<int> status(OFX_OK);
    const int n_workers = n_ctx < n_groups ? n_ctx : n_groups;
    std::vector<int> conc(n_ctx);
    for (int w = 0; w < n_ctx; w++) {                           // memory budgets are shared between the workers
        conc[w] = ctxs[w]->concurrency;
        if (ctxs[w]->concurrency < n_workers) ctxs[w]->concurrency = n_workers;
    }
    auto worker = [&](int w) {
        std::vector<ofx_stats> st(G);
        for (int q = w; q < n_groups; q += n_ctx) {
            if (status.load() != OFX_OK) return;
            const int first = q * G, cnt = (n_pairs - first < G) ? n_pairs - first : G;
            const int s = group(ctxs[w], first, cnt, st.data());
            if (s != OFX_OK) { int expected = OFX_OK; status.compare_exchange_strong(expected, s); return; }
            if (work_pix_iters)
                for (int g = 0; g < cnt; g++) work_pix_iters[first + g] = st[g].work_pix_iters;
        }
        (void) hipStreamSynchronize(ctxs[w]->stream);
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_ctx && w < n_groups; w++) th.emplace_back(worker, w);
    worker(0);
    for (auto &t : th) t.join();
    for (int w = 0; w < n_ctx; w++) ctxs[w]->concurrency = conc[w];
    return status.load();
}

// group size of the SOR batches: option "lockstep" of ctxs[0], else as large as possible (16), evened out over the
// contexts like ofx_tvl1_batch_group_size
static int sor_batch_group_size(ofx_ctx *const *ctxs, int n_ctx, int n_pairs)
{
    int G = ctxs[0]->lockstep;
    if (G > 0) return G > OFX_MAX_GROUP ? OFX_MAX_GROUP : G;
    if (n_pairs <= 1) return 1;
    const int cap = OFX_MAX_GROUP;
    const int rounds = (n_pairs + n_ctx * cap - 1) / (n_ctx * cap);
    G = (n_pairs + n_ctx * rounds - 1) / (n_ctx * rounds);
    return G < 1 ? 1 : G;
}

static int sor_batch_check(ofx_ctx *const *ctxs, int n_ctx, int n_pairs, const void *a, const void *b, const void *c)
{
    if (!ctxs || n_ctx < 1 || n_pairs < 0 || !a || !b || !c) return OFX_ERR_ARG;
    for (int w = 0; w < n_ctx; w++)
        if (!ctxs[w] || ctxs[w]->device != ctxs[0]->device || ctxs[w]->precision != ctxs[0]->precision) return OFX_ERR_ARG;
    return OFX_OK;
}

extern "C" int ofx_hs_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI1, const void *const *dI2,
                                void *const *d_flo, int n_pairs, int nx, int ny, double alpha, int nscales, double zfactor,
                                int warps, double TOL, int maxiter, double *work_pix_iters)
{
    OFX_TRY(sor_batch_check(ctxs, n_ctx, n_pairs, dI1, dI2, d_flo));
    if (n_pairs == 0) return OFX_OK;
    const int G = sor_batch_group_size(ctxs, n_ctx, n_pairs);
    return sor_batch_run(ctxs, n_ctx, n_pairs, G, work_pix_iters, [&](ofx_ctx *c, int first, int cnt, ofx_stats *st) {
        return ofx_hs_group_dev(c, cnt, dI1 + first, dI2 + first, d_flo + first, nx, ny, alpha, nscales, zfactor, warps, TOL,
                                maxiter, st);
    });
}

extern "C" int ofx_brox_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI1, const void *const *dI2,
                                  void *const *d_flo, int n_pairs, int nxx, int nyy, double alpha, double gamma, int nscales,
                                  double nu, double TOL, int inner_iter, int outer_iter, double *work_pix_iters)
{
    OFX_TRY(sor_batch_check(ctxs, n_ctx, n_pairs, dI1, dI2, d_flo));
    if (n_pairs == 0) return OFX_OK;
    const int G = sor_batch_group_size(ctxs, n_ctx, n_pairs);
    return sor_batch_run(ctxs, n_ctx, n_pairs, G, work_pix_iters, [&](ofx_ctx *c, int first, int cnt, ofx_stats *st) {
        return ofx_brox_group_dev(c, cnt, dI1 + first, dI2 + first, d_flo + first, nxx, nyy, alpha, gamma, nscales, nu, TOL,
                                  inner_iter, outer_iter, st);
    });
}


// ============================================================================================
// Brox temporal: src/brox_optic_flow_temporal.cpp, src/brox_temporal_mask.cpp
// ============================================================================================
// A sequence of `frames` images gives nz = frames - 1 flow fields, coupled by a temporal smoothness term.  All
// arrays are frame-major (frame f at element f * nx * ny); flow field f uses I1 = frame f, I2 = frame f + 1, so the
// per-frame kernels of the spatial method (prepare, warp, coefficient assembly) are reused unchanged.  New: the
// smoothness weight with the temporal flow derivative, psi5 / psi6 and their divergence terms, the 7-point SOR
// update, and the frame dimension in the exact schedule (a sweep visits frames 1 .. nz-2, then 0, then nz-1,
// :439-459; every frame in the spatial method's pixel order).

struct Psi6 { double p1, p2, p3, p4, p5, p6; };
// src/brox_temporal_mask.cpp:18-132 at one pixel: psi1..4 within the frame, psi5 / psi6 = half-sums with the previous /
// following frame (0 outside the sequence)
template <typename T> OFX_DEV Psi6 broxt_psi6(const T *Psis, int f, int i, int j, int nx, int ny, int nz)
{
    const size_t df = (size_t) nx * ny;
    const Psi4 a = brox_psi4(Psis + f * df, i, j, nx, ny);
    const size_t k = f * df + (size_t) i * nx + j;
    const double c = ldw(Psis + k);
    Psi6 r;
    r.p1 = a.p1; r.p2 = a.p2; r.p3 = a.p3; r.p4 = a.p4;
    r.p5 = (f > 0) ? 0.5 * (ldw(Psis + k - df) + c) : 0.0;
    r.p6 = (f < nz - 1) ? 0.5 * (ldw(Psis + k + df) + c) : 0.0;
    return r;
}

// psi_smooth (:94-118) of centered_gradient3 (src/operators.cpp:413-499): the in-frame centred gradient plus the
// temporal centred difference (one-sided, still x 0.5, in the first / last frame)
template <typename T>
__global__ void k_broxt_psis(const typename Pix<T>::v2 *__restrict__ U, T *__restrict__ Psis, int nx, int ny, int nz)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (j >= nx || i >= ny) return;
    const size_t df = (size_t) nx * ny;
    const typename Pix<T>::v2 *Uf = U + f * df;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const size_t p = (size_t) i * nx + j;
    const double2 r = ldw2(Uf + (size_t) i * nx + jr), l = ldw2(Uf + (size_t) i * nx + jl);
    const double2 d = ldw2(Uf + (size_t) id * nx + j), t = ldw2(Uf + (size_t) iu * nx + j);
    const double2 hi = ldw2(U + (f < nz - 1 ? f + 1 : f) * df + p), lo = ldw2(U + (f > 0 ? f - 1 : f) * df + p);
    const double ux = 0.5 * (r.x - l.x), uy = 0.5 * (d.x - t.x), ut = 0.5 * (hi.x - lo.x);
    const double vx = 0.5 * (r.y - l.y), vy = 0.5 * (d.y - t.y), vt = 0.5 * (hi.y - lo.y);
    const double du = ux * ux + uy * uy + ut * ut;
    const double dv = vx * vx + vy * vy + vt * vt;
    const double d2 = du + dv;
    stn(Psis + f * df + p, 1. / sqrt(d2 + BROX_EPSILON * BROX_EPSILON));
}

// div_u, div_v (src/brox_temporal_mask.cpp:140-239: the in-frame sum, then `+=` the temporal terms as ONE added
// expression), div_d = alpha (psi1 + ... + psi6) and du = dv = 0 (:383-390)
template <typename T>
__global__ void k_broxt_div(const typename Pix<T>::v2 *__restrict__ U, const T *__restrict__ Psis,
                            typename Pix<T>::v2 *__restrict__ DV, T *__restrict__ Dd, typename Pix<T>::v2 *__restrict__ DU,
                            int nx, int ny, int nz, double alpha)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (j >= nx || i >= ny) return;
    const size_t df = (size_t) nx * ny;
    const size_t k = f * df + (size_t) i * nx + j;
    const Psi6 s = broxt_psi6(Psis, f, i, j, nx, ny, nz);
    const double2 c = ldw2(U + k);
    double du = 0.0, dv = 0.0;
    bool have = false;
    if (i < ny - 1) { const double2 q = ldw2(U + k + nx); du = s.p1 * (q.x - c.x); dv = s.p1 * (q.y - c.y); have = true; }
    if (i > 0) {
        const double2 q = ldw2(U + k - nx);
        const double a = s.p2 * (q.x - c.x), b = s.p2 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    if (j < nx - 1) {
        const double2 q = ldw2(U + k + 1);
        const double a = s.p3 * (q.x - c.x), b = s.p3 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    if (j > 0) {
        const double2 q = ldw2(U + k - 1);
        const double a = s.p4 * (q.x - c.x), b = s.p4 * (q.y - c.y);
        du = have ? du + a : a; dv = have ? dv + b : b; have = true;
    }
    if (nz > 1) {
        if (f > 0 && f < nz - 1) {
            const double2 lo = ldw2(U + k - df), hi = ldw2(U + k + df);
            du += s.p5 * (lo.x - c.x) + s.p6 * (hi.x - c.x);
            dv += s.p5 * (lo.y - c.y) + s.p6 * (hi.y - c.y);
        } else if (f == 0) {
            const double2 hi = ldw2(U + k + df);
            du += s.p6 * (hi.x - c.x);
            dv += s.p6 * (hi.y - c.y);
        } else {
            const double2 lo = ldw2(U + k - df);
            du += s.p5 * (lo.x - c.x);
            dv += s.p5 * (lo.y - c.y);
        }
    }
    stn2(DV + k, make_double2(du, dv));
    stn(Dd + k, alpha * (s.p1 + s.p2 + s.p3 + s.p4 + s.p5 + s.p6));
    stn2(DU + k, make_double2(0.0, 0.0));
}

// SOR update of one pixel of frame f, :120-170; a missing neighbour (row, column or frame) is the pixel itself with
// psi = 0.  The new value also goes into the sweep's snapshot.
template <typename T>
OFX_DEV double broxt_point(typename Pix<T>::v2 *DU, typename Pix<T>::v2 *snap, const typename Pix<T>::v4 *__restrict__ CO,
                           const T *__restrict__ Dm, const T *__restrict__ Psis, int f, int i, int j, int nx, int ny, int nz,
                           double alpha)
{
    const size_t df = (size_t) nx * ny;
    const size_t k = f * df + (size_t) i * nx + j;
    const Psi6 s = broxt_psi6(Psis, f, i, j, nx, ny, nz);
    const double2 c = ldw2(DU + k);
    const double2 dn = (i < ny - 1) ? ldw2(DU + k + nx) : c, up = (i > 0) ? ldw2(DU + k - nx) : c;
    const double2 rt = (j < nx - 1) ? ldw2(DU + k + 1) : c, lf = (j > 0) ? ldw2(DU + k - 1) : c;
    const double2 pv = (f > 0) ? ldw2(DU + k - df) : c, nxt = (f < nz - 1) ? ldw2(DU + k + df) : c;
    const double4 co = ldw4(CO + k);
    const double D = ldw(Dm + k);
    const double w = BROX_SOR_W;
    const double div_du = s.p1 * dn.x + s.p2 * up.x + s.p3 * rt.x + s.p4 * lf.x + s.p5 * pv.x + s.p6 * nxt.x;      // :157-159
    const double div_dv = s.p1 * dn.y + s.p2 * up.y + s.p3 * rt.y + s.p4 * lf.y + s.p5 * pv.y + s.p6 * nxt.y;      // :160-162
    const double duk = c.x, dvk = c.y;
    const double dun = rnd_to<T>((1. - w) * duk + w * (co.x - D * dvk + alpha * div_du) / co.z);   // :167
    const double dvn = rnd_to<T>((1. - w) * dvk + w * (co.y - D * dun + alpha * div_dv) / co.w);   // :168
    stn2(DU + k, make_double2(dun, dvn));
    stn2(snap + k, make_double2(dun, dvn));
    return (dun - duk) * (dun - duk) + (dvn - dvk) * (dvn - dvk);                                  // :171
}

// Windowed exact schedule (k_brox_window) with the frame dimension: one workgroup per (sweep, frame, row block).
// blockIdx.y = position o of the frame in the sweep's visiting order: frames 1 .. nz-2, then 0, then nz-1 (:439-459).
template <typename T>
__global__ __launch_bounds__(1024) void k_broxt_window(typename Pix<T>::v2 *DU, typename Pix<T>::v2 *snap,
                                                       const typename Pix<T>::v4 *__restrict__ CO, const T *__restrict__ Dm,
                                                       const T *__restrict__ Psis, double *__restrict__ err, SorWin w,
                                                       int nx, int ny, int nz, double alpha)
{
    const int b = blockIdx.x, o = blockIdx.y, s = w.s_first + blockIdx.z;
    const int f = (o < nz - 2) ? o + 1 : (o == nz - 2 ? 0 : nz - 1);
    const int qmax = ny + nx - 2;
    const int q_first = w.tau0 - w.lag_s * s - w.lag_f * o - w.lag_b * b;
    if (sor_unit_idle(w, b, q_first, w.K, 1, nx, ny)) return;
    typename Pix<T>::v2 *mysnap = snap + (size_t) s * nz * nx * ny;
    const int r = sor_window_item(w, b, threadIdx.x, ny);
    double e = 0.0;
    for (int q = q_first; q < q_first + w.K; q++) {
        if (q >= 0 && q <= qmax && r >= 0) {
            int i, j;
            if (r == ny + 2) {
                for (int corner = 0; corner < 4; corner++)
                    if (brox_plane_item(r, q, nx, ny, corner, i, j) && sor_border_block(i, ny, w.R) == b)
                        e += broxt_point<T>(DU, mysnap, CO, Dm, Psis, f, i, j, nx, ny, nz, alpha);
            } else if (brox_plane_item(r, q, nx, ny, 0, i, j) && (r < ny || sor_border_block(i, ny, w.R) == b)) {
                e += broxt_point<T>(DU, mysnap, CO, Dm, Psis, f, i, j, nx, ny, nz, alpha);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    loop_accumulate(err, s, e, (o * 8 + b) * 4 + (threadIdx.x >> 6));
}

template <typename T> struct BroxtLevel {
    using v2 = typename Pix<T>::v2;
    using v4 = typename Pix<T>::v4;
    int nx, ny;
    T *I;                       // frames * n
    T *Psis, *Dd, *Dm;          // nz * n each
    v2 *G1, *PB, *WB, *U, *DV, *DU, *Snap;
    v4 *PA, *WA, *CO;
    int snap_planes, sweep_hint;
};

template <typename T> static int broxt_level_alloc(ofx_ctx *ctx, BroxtLevel<T> &L, int nx, int ny, int frames)
{
    const size_t n1 = (size_t) nx * ny * (frames - 1);
    L.nx = nx;
    L.ny = ny;
    OFX_TRY(ofx_alloc(ctx, (size_t) nx * ny * frames, &L.I));
    OFX_TRY(ofx_alloc(ctx, n1, &L.Psis));
    OFX_TRY(ofx_alloc(ctx, n1, &L.Dd));
    OFX_TRY(ofx_alloc(ctx, n1, &L.Dm));
    OFX_TRY(ofx_alloc(ctx, n1, &L.G1));
    OFX_TRY(ofx_alloc(ctx, n1, &L.PB));
    OFX_TRY(ofx_alloc(ctx, n1, &L.WB));
    OFX_TRY(ofx_alloc(ctx, n1, &L.U));
    OFX_TRY(ofx_alloc(ctx, n1, &L.DV));
    OFX_TRY(ofx_alloc(ctx, n1, &L.DU));
    OFX_TRY(ofx_alloc(ctx, n1, &L.PA));
    OFX_TRY(ofx_alloc(ctx, n1, &L.WA));
    OFX_TRY(ofx_alloc(ctx, n1, &L.CO));
    L.Snap = nullptr;
    L.snap_planes = 0;
    L.sweep_hint = 0;
    return OFX_OK;
}

// src/brox_optic_flow_temporal.cpp:282-512 on device data
template <typename T>
static int broxt_single_scale_dev(ofx_ctx *ctx, BroxtLevel<T> &L, const BroxParams &P, int frames, int scale)
{
    const int nx = L.nx, ny = L.ny, nz = frames - 1, n = nx * ny;
    const size_t n1 = (size_t) n * nz;
    if ((long long) n1 >= (1LL << 31)) return ofx_fail(ctx, OFX_ERR_ARG, "brox temporal: sequence larger than 2^31 pixels");
    if (nx < 3 || ny < 3) return ofx_fail(ctx, OFX_ERR_ARG, "brox temporal: level %dx%d has no interior", nx, ny);
    const dim3 g = g2d(nx, ny), b = b2d(), g3(g.x, g.y, nz);
    const dim3 g1((unsigned) ((n1 + 255) / 256)), b1(256);
    ofx_stats &S = ctx->stats;
    int solve = 0;
    for (int f = 0; f < nz; f++) {                                                                   // :346-355
        hipLaunchKernelGGL(k_brox_prepare<T>, g, b, 0, ctx->stream, (const T *) (L.I + (size_t) f * n),
                           (const T *) (L.I + (size_t) (f + 1) * n), L.G1 + (size_t) f * n, L.PA + (size_t) f * n,
                           L.PB + (size_t) f * n, nx, ny);
    }
    OFX_LAUNCH_CHECK(ctx);
    for (int no = 0; no < P.outer_iter; no++) {                                                      // :358
        for (int f = 0; f < nz; f++)                                                                 // :360-367
            hipLaunchKernelGGL(k_brox_warp<T>, g, b, 0, ctx->stream, L.PA + (size_t) f * n, L.PB + (size_t) f * n,
                               L.U + (size_t) f * n, L.WA + (size_t) f * n, L.WB + (size_t) f * n, nx, ny);
        hipLaunchKernelGGL(k_broxt_psis<T>, g3, b, 0, ctx->stream, L.U, L.Psis, nx, ny, nz);          // :370-374
        hipLaunchKernelGGL(k_broxt_div<T>, g3, b, 0, ctx->stream, L.U, (const T *) L.Psis, L.DV, L.Dd, L.DU, nx, ny, nz,
                           P.alpha);                                                                 // :377-390
        OFX_LAUNCH_CHECK(ctx);
        for (int ni = 0; ni < P.inner_iter; ni++) {                                                  // :394
            hipLaunchKernelGGL(k_brox_coeff<T>, g1, b1, 0, ctx->stream, (const T *) L.I, L.G1, L.WA, L.WB, L.DU, L.DV,
                               (const T *) L.Dd, L.CO, L.Dm, (int) n1, P.alpha, P.gamma);            // :396-427
            OFX_LAUNCH_CHECK(ctx);
            int nsor = 0;
            double error = 1000;
            const size_t ub = n1 * sizeof(typename Pix<T>::v2);
            const int batch = sor_pick_batch(ctx, n1, sizeof(typename Pix<T>::v2), OFX_BROX_MAX_ITERATIONS);
            if (L.snap_planes < batch) {
                OFX_TRY(ofx_alloc(ctx, n1 * batch, &L.Snap));
                L.snap_planes = batch;
            }
            auto window = [&](const SorWin &w, int blocks, int sweeps, unsigned, int) -> int {
                hipLaunchKernelGGL(k_broxt_window<T>, dim3(blocks, nz, sweeps), dim3(sor_window_threads(w.R + 3)), 0,
                                   ctx->stream, L.DU, L.Snap, L.CO, (const T *) L.Dm, (const T *) L.Psis, ctx->d_err, w, nx,
                                   ny, nz, P.alpha);
                OFX_LAUNCH_CHECK(ctx);
                return OFX_OK;
            };
            auto take = [&](int, int k) -> int {
                OFX_HIP(ctx, hipMemcpyAsync(L.DU, L.Snap + (size_t) (k - 1) * n1, ub, hipMemcpyDeviceToDevice, ctx->stream));
                return OFX_OK;
            };
            OFX_TRY(sor_window_loop(ctx, 1, (int) n1, ny, P.TOL, OFX_BROX_MAX_ITERATIONS, ny + nx - 2, BROX_PLANE_C, batch, window,
                                    take, &nsor, &error, nz, &L.sweep_hint));                                       // :430-461
            if (P.verbose) { printf("Iterations: %d\n", nsor); fflush(stdout); }                     // :463-465
            if (scale < OFX_MAX_SCALES) {
                if (solve < OFX_MAX_SOLVES) { S.iters[scale][solve] = nsor; S.error[scale][solve] = error; }
                S.iter_launches[scale] += nsor;
            }
            S.work_pix_iters += (double) nsor * n1;
            solve++;
        }
        hipLaunchKernelGGL(k_brox_add<T>, g1, b1, 0, ctx->stream, L.U, L.DU, (int) n1);                // :469-472
        OFX_LAUNCH_CHECK(ctx);
    }
    return OFX_OK;
}

// src/brox_optic_flow_temporal.cpp:520-627
template <typename T>
static int broxt_host(ofx_ctx *ctx, const double *I, double *u, double *v, int nx, int ny, int frames, const BroxParams &P,
                      int nscales, double nu)
{
    const size_t n = (size_t) nx * ny;
    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nx, ny, nscales, nu, nxs, nys));
    sor_stats_begin(ctx, nscales, P.inner_iter * P.outer_iter);
    T *dI, *dummy, *tmpA, *tmpB;
    double *scr;
    OFX_TRY(upload_plane<T>(ctx, I, n * frames, &dI));
    OFX_TRY(ofx_alloc(ctx, n * frames, &dummy));
    OFX_TRY(ofx_alloc(ctx, n, &tmpA));
    OFX_TRY(ofx_alloc(ctx, n, &tmpB));
    OFX_TRY(ofx_alloc(ctx, op_pyramid_scratch_doubles(), &scr));
    std::vector<BroxtLevel<T>> lv(nscales);
    for (int s = 0; s < nscales; s++) {
        OFX_TRY(broxt_level_alloc<T>(ctx, lv[s], nxs[s], nys[s], frames));
        if (s < OFX_MAX_SCALES) { ctx->stats.nx[s] = nxs[s]; ctx->stats.ny[s] = nys[s]; }
    }
    // image_normalization_1 over the whole sequence (:548): the joint-min/max kernel of normalization_2 fed the
    // sequence twice computes the same 255 (I - min) / den
    OFX_TRY(op_normalize2<T>(ctx, dI, dI, lv[0].I, dummy, (int) (n * frames), scr));
    for (int f = 0; f < frames; f++) OFX_TRY(op_gaussian<T>(ctx, lv[0].I + f * n, tmpA, nx, ny, BROX_SIGMA));     // :551-553
    for (int s = 1; s < nscales; s++) {                                                              // :561-575
        const size_t np = (size_t) nxs[s - 1] * nys[s - 1], nc = (size_t) nxs[s] * nys[s];
        for (int f = 0; f < frames; f++)
            OFX_TRY(op_zoom_out<T>(ctx, lv[s - 1].I + f * np, lv[s].I + f * nc, tmpA, tmpB, nxs[s - 1], nys[s - 1], nu));
    }
    BroxtLevel<T> &C = lv[nscales - 1];
    OFX_TRY(op_fill2<T>(ctx, C.U, (size_t) C.nx * C.ny * (frames - 1)));                             // :578-580
    for (int s = nscales - 1; s >= 0; s--) {                                                        // :587
        if (P.verbose) { printf("Scale: %d\n", s); fflush(stdout); }
        OFX_TRY(broxt_single_scale_dev<T>(ctx, lv[s], P, frames, s));
        if (s) {
            const size_t nc = (size_t) lv[s].nx * lv[s].ny, nf = (size_t) lv[s - 1].nx * lv[s - 1].ny;
            for (int f = 0; f < frames - 1; f++)
                OFX_TRY(op_zoom_in_flow<T>(ctx, lv[s].U + f * nc, lv[s - 1].U + f * nf, lv[s].nx, lv[s].ny, lv[s - 1].nx,
                                           lv[s - 1].ny, 1.0 / nu));                                 // :600-612
        }
    }
    return download_flow<T>(ctx, lv[0].U, u, v, n * (frames - 1));
}

extern "C" int ofx_brox_temporal(ofx_ctx *ctx, const double *I, double *u, double *v, int nxx, int nyy, int frames,
                                 double alpha, double gamma, int nscales, double nu, double TOL, int inner_iter,
                                 int outer_iter, int verbose)
{
    OFX_ENTER(ctx);
    if (!I || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "brox temporal: NULL pointer");
    if (frames <= 2) return ofx_fail(ctx, OFX_ERR_ARG, "The method needs more than two frames");     // :537-541
    if (inner_iter < 0 || outer_iter < 0) return ofx_fail(ctx, OFX_ERR_ARG, "brox temporal: negative iteration count");
    if (nxx < 3 || nyy < 3) return ofx_fail(ctx, OFX_ERR_ARG, "brox temporal: images smaller than 3x3");
    const double t0 = ofx_now_ms();
    const BroxParams P = {alpha, gamma, TOL, inner_iter, outer_iter, verbose};
    int s = ctx->precision == OFX_F64 ? broxt_host<double>(ctx, I, u, v, nxx, nyy, frames, P, nscales, nu)
                                      : broxt_host<float>(ctx, I, u, v, nxx, nyy, frames, P, nscales, nu);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}


// ============================================================================================
// Classic Horn-Schunck: src/horn_schunck_classic.cpp -- n Jacobi iterations from a zero flow
// ============================================================================================
// Jacobi, so exactly parallel: one launch per iteration on ping-pong (u, v) pairs.  Per pixel and iteration:
// read (Ex, Ey) + Et (3 T) and the 3x3 neighbourhood of (u, v) (2 T compulsory), write (u, v) (2 T) = 7 T = 56 B
// in f64 -- HBM-bound.  Sample p(x, i, j) = column i, row j with clamped indices (:22-43).
template <typename T> OFX_DEV double hsc_p(const T *x, int w, int h, int i, int j)
{
    i = i < 0 ? 0 : (i >= w ? w - 1 : i);
    j = j < 0 ? 0 : (j >= h ? h - 1 : j);
    return ldw(x + (size_t) j * w + i);
}

// compute_input_derivatives, :46-73 (every sum in the reference's left-to-right order)
template <typename T>
__global__ void k_hsc_derivs(const T *__restrict__ a, const T *__restrict__ b, typename Pix<T>::v2 *__restrict__ E,
                             T *__restrict__ Et, int w, int h)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (i >= w || j >= h) return;
    const double a00 = hsc_p(a, w, h, i, j), a10 = hsc_p(a, w, h, i + 1, j), a01 = hsc_p(a, w, h, i, j + 1),
                 a11 = hsc_p(a, w, h, i + 1, j + 1);
    const double b00 = hsc_p(b, w, h, i, j), b10 = hsc_p(b, w, h, i + 1, j), b01 = hsc_p(b, w, h, i, j + 1),
                 b11 = hsc_p(b, w, h, i + 1, j + 1);
    const double ey = (1.0 / 4) * (a01 - a00 + a11 - a10 + b01 - b00 + b11 - b10);
    const double ex = (1.0 / 4) * (a10 - a00 + a11 - a01 + b10 - b00 + b11 - b01);
    const double et = (1.0 / 4) * (b00 - a00 + b10 - a10 + b01 - a01 + b11 - a11);
    const size_t p = (size_t) j * w + i;
    stn2(E + p, make_double2(ex, ey));
    stn(Et + p, et);
}

// compute_bar + hs_iteration, :76-122
template <typename T>
__global__ __launch_bounds__(256) void k_hsc_iter(const typename Pix<T>::v2 *__restrict__ Uin,
                                                  typename Pix<T>::v2 *__restrict__ Uout,
                                                  const typename Pix<T>::v2 *__restrict__ E, const T *__restrict__ Et, int w,
                                                  int h, double alpha)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (i >= w || j >= h) return;
    const int il = i > 0 ? i - 1 : 0, ir = i < w - 1 ? i + 1 : w - 1;
    const unsigned rc = (unsigned) j * w, ru = (unsigned) (j > 0 ? j - 1 : 0) * w, rd = (unsigned) (j < h - 1 ? j + 1 : h - 1) * w;
    const double2 l = ldw2(Uin + rc + il), r = ldw2(Uin + rc + ir), t = ldw2(Uin + ru + i), d = ldw2(Uin + rd + i);
    const double2 tl = ldw2(Uin + ru + il), tr = ldw2(Uin + ru + ir), dl = ldw2(Uin + rd + il), dr = ldw2(Uin + rd + ir);
    const double ubar = (1.0 / 6) * (l.x + r.x + t.x + d.x) + (1.0 / 12) * (tl.x + tr.x + dl.x + dr.x);
    const double vbar = (1.0 / 6) * (l.y + r.y + t.y + d.y) + (1.0 / 12) * (tl.y + tr.y + dl.y + dr.y);
    const double2 e = ldw2(E + rc + i);
    const double et = ldw(Et + rc + i);
    double tt = e.x * ubar + e.y * vbar + et;
    tt /= alpha * alpha + e.x * e.x + e.y * e.y;
    stn2(Uout + rc + i, make_double2(rnd_to<T>(ubar - e.x * tt), rnd_to<T>(vbar - e.y * tt)));
}

template <typename T>
static int hs_classic_host(ofx_ctx *ctx, const double *a, const double *b, double *u, double *v, int w, int h, int niter,
                           double alpha)
{
    const size_t n = (size_t) w * h;
    T *da, *db, *Et;
    typename Pix<T>::v2 *U[2], *E;
    OFX_TRY(upload_plane<T>(ctx, a, n, &da));
    OFX_TRY(upload_plane<T>(ctx, b, n, &db));
    OFX_TRY(ofx_alloc(ctx, n, &U[0]));
    OFX_TRY(ofx_alloc(ctx, n, &U[1]));
    OFX_TRY(ofx_alloc(ctx, n, &E));
    OFX_TRY(ofx_alloc(ctx, n, &Et));
    hipLaunchKernelGGL(k_hsc_derivs<T>, g2d(w, h), b2d(), 0, ctx->stream, (const T *) da, (const T *) db, E, Et, w, h);
    OFX_LAUNCH_CHECK(ctx);
    OFX_TRY(op_fill2<T>(ctx, U[0], n));                                                            // :139-141
    int cur = 0;
    for (int it = 0; it < niter; it++) {                                                           // :142-144
        hipLaunchKernelGGL(k_hsc_iter<T>, g2d(w, h), b2d(), 0, ctx->stream, U[cur], U[cur ^ 1], E, (const T *) Et, w, h,
                           alpha);
        cur ^= 1;
    }
    OFX_LAUNCH_CHECK(ctx);
    sor_stats_begin(ctx, 1, 1);
    ctx->stats.nx[0] = w;
    ctx->stats.ny[0] = h;
    ctx->stats.iters[0][0] = niter;
    ctx->stats.work_pix_iters = (double) niter * n;
    return download_flow<T>(ctx, U[cur], u, v, n);
}

extern "C" int ofx_hs_classic(ofx_ctx *ctx, const double *a, const double *b, double *u, double *v, int w, int h, int niter,
                              double alpha)
{
    OFX_ENTER(ctx);
    if (!a || !b || !u || !v) return ofx_fail(ctx, OFX_ERR_ARG, "hs_classic: NULL pointer");
    if (w < 1 || h < 1 || (long long) w * h >= (1LL << 28)) return ofx_fail(ctx, OFX_ERR_ARG, "hs_classic: bad size %dx%d", w, h);
    if (niter < 0) niter = 0;                                   // the reference's loop simply does not run
    const double t0 = ofx_now_ms();
    int s = ctx->precision == OFX_F64 ? hs_classic_host<double>(ctx, a, b, u, v, w, h, niter, alpha)
                                      : hs_classic_host<float>(ctx, a, b, u, v, w, h, niter, alpha);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}
