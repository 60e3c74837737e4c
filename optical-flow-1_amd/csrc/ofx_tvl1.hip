// ofx_tvl1.hip -- TV-L1 primal-dual solver on gfx950 (reference: src/tvl1flow.cpp).
//
// Device data layout of one pyramid level (T = storage type, N = nx*ny, row-major):
//   U[2]   : (u1,u2)        interleaved pairs, ping-pong          2 T / px each
//   P1[2]  : (p11,p12)      interleaved pairs, ping-pong          2 T / px each
//   P2[2]  : (p21,p22)      interleaved pairs, ping-pong          2 T / px each
//   A      : (I1wx,I1wy)    warped gradient, constant per warp    2 T / px
//   R      : rho_c          constant part of rho, per warp        1 T / px
//   pa, pb : (I1,I1x), I1y  target image + centred gradient       3 T / px (gathered by the warp)
//   I0     : source image                                          1 T / px
// Every stream is read/written as whole 16-byte (f64) pairs per lane with unit stride, i.e. each
// wave-instruction moves one contiguous 1 KiB segment.
//
// k_tvl1_iter = ONE launch per inner iteration of src/tvl1flow.cpp:113-182, fusing all five sweeps
// (threshold, div p, u update + error, grad u, dual update).  Compulsory traffic per pixel and
// iteration: read U,P1,P2,A,R (9 T) + write U,P1,P2 (6 T) = 15 T = 120 B (f64) / 60 B (f32); `grad`
// is recomputed from A (bit-identical in f64).  The as-written reference moves 40 T.
//
// Work decomposition: one WAVE owns a strip of 62 output columns (64 lanes = 62 + one halo column on
// each side) and marches down `rows` image rows.  Vertical neighbours (p12/p22 of the row above,
// u_new of the row below) live in the lane's own registers across marching steps; horizontal
// neighbours (p11/p21 of the left column, u_new of the right column) come from the adjacent lane
// through a wave shift.  No LDS, no barriers, no inter-wave communication; the only redundant work
// is the two halo lanes (3 %) and one extra row per strip (u_new of the row below the strip).
// u and p are ping-pong buffered because neighbouring strips read this strip's old values.
//
// Convergence test without per-iteration host sync: each wave adds its partial sum of |du|^2 into
// err[k][wave % 64] (one f64 atomic per wave); the NEXT launch begins by summing the 64 shards of
// err[k-1] in a fixed order and returns immediately if `error > eps^2` is false -- once one iteration
// is "done" every later launch sees an untouched (zero) slot and is a no-op too.  The host enqueues
// iterations in chunks, each followed by a one-block finalize kernel that publishes {n, done, error};
// it polls that record one chunk behind, so the GPU never idles and `n` is exactly the reference's.
#include "ofx_ops.h"
#include "ofx_device.h"
#include "ofx_loop.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

#define TVL1_GRAD_IS_ZERO 1E-10          // src/tvl1flow.cpp:24
#define TVL1_PRESMOOTHING_SIGMA 0.8      // src/tvl1flow.cpp:23
#define STRIP_OUT 62                     // output columns per wave

template <typename T> OFX_DEV double rnd_to(double x);
template <> OFX_DEV double rnd_to<double>(double x) { return x; }
template <> OFX_DEV double rnd_to<float>(double x) { return (double) (float) x; }

// The wave index of the marching kernels: the same in every lane (256-thread blocks of four waves), but derived from threadIdx,
// so the compiler takes it -- and the strip geometry, the row counter and every stage condition computed from it -- for
// lane-varying: vector compares, exec-mask branches around every pipeline stage, the loop bounds in VGPRs.  readfirstlane
// states the uniformity: scalar compares and branches, 13-15 fewer VGPRs.  (OFX_NO_RFL: the A/B build without it.)
#ifdef OFX_NO_RFL
#define OFX_WAVE_UNIFORM(x) (x)
#else
#define OFX_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
template <typename T> struct RowIn {
    double2 u, p1, p2, a;
    double r;
};

// All streams of a level are addressed as  uniform base pointer + 32-bit per-lane BYTE offset, which is
// the addressing mode of global_load/store (SGPR base + zero-extended VGPR offset): one v_add per row
// step instead of 64-bit index arithmetic per access.  Requires nx*ny*16 < 2^32 (checked by the host).
template <typename V> OFX_DEV const V *at(const V *base, unsigned byte_off)
{
    return reinterpret_cast<const V *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename V> OFX_DEV V *at(V *base, unsigned byte_off)
{
    return reinterpret_cast<V *>(reinterpret_cast<char *>(base) + byte_off);
}

// off2 = byte offset of the pixel in the pair arrays (2 T per pixel); rho_c (1 T per pixel) is at off2 / 2
template <typename T>
OFX_DEV RowIn<T> tvl1_load_row(const typename Pix<T>::v2 *U, const typename Pix<T>::v2 *P1,
                               const typename Pix<T>::v2 *P2, const typename Pix<T>::v2 *A, const T *R, unsigned off2)
{
    RowIn<T> r;
    r.u = ldw2(at(U, off2));
    r.p1 = ldw2(at(P1, off2));
    r.p2 = ldw2(at(P2, off2));
    r.a = ldw2(at(A, off2));
    r.r = ldw(at(R, off2 >> 1));
    return r;
}

// The tolerance / f32 arithmetic of the TV-L1 stages (not the strict one): ONE refinement step on the 2^-23 estimates of
// v_rsq_f64 / v_rcp_f64 -- relative error ~2^-45, eleven orders of magnitude inside the AEPE bar -- instead of the two steps
// + residual correction that make the last bit right.  Issue cost (tools/ubench/f64_issue.hip: a plain f64 instruction is 4
// cycles per wave, v_rcp / v_rsq / v_sqrt_f64 16): square root 4 + 4 against 4 + 9 units, reciprocal 4 + 2 against 4 + 4.
#ifndef OFX_TOL_NEWTON
#define OFX_TOL_NEWTON 1
#endif
OFX_DEV double sqrt_tol(double x)
{
#if OFX_TOL_NEWTON >= 2
    return sqrt_unscaled(x);
#else
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
#endif
}
OFX_DEV double rcp_tol(double d)
{
#if OFX_TOL_NEWTON >= 2
    return rcp_newton(d);
#else
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
#endif
}

// Stage "primal" at one pixel: thresholding TH (src/tvl1flow.cpp:117-143), divergence of p
// (src/operators.cpp:35-78) and u = v + theta div p (:156-157).  l11/l21 = p11/p21 of the left pixel,
// up12/up22 = p12/p22 of the pixel above.  Returns the new (u1,u2), rounded to the storage type.
template <typename T, bool STRICT = true>
OFX_DEV double2 tvl1_primal(double2 u, double2 a, double r, double2 p1, double2 p2, double l11, double l21, double up12,
                            double up22, bool lef, bool rig, bool top, bool bot, double l_t, double theta)
{
#ifdef OFX_CEIL_MEM   // ceiling experiment (tools/ceilings.sh): memory traffic and lane shifts kept, arithmetic replaced by a copy
    return make_double2(u.x + 1e-300 * (a.x + r + p1.x + l11 + up12), u.y + 1e-300 * (a.y + p2.x + p1.y + p2.y + l21 + up22));
#endif
    // (A variant with the interior divergence behind a wave-uniform test and the thresholding as selects -- fewer, longer basic
    // blocks -- measured the same, profiles/r03_v_ab_wave_uniform_index.txt: the stalls of these kernels are the latencies of
    // the dependent f64 chains, not their branches.)
    const double div1 = div_backward(p1.x, l11, p1.y, up12, lef, rig, top, bot);
    const double div2 = div_backward(p2.x, l21, p2.y, up22, lef, rig, top, bot);
    const double ix = a.x, iy = a.y;
    const double grad = ix * ix + iy * iy;                  // :100-104
    const double rho = r + (ix * u.x + iy * u.y);           // :119-120
    const double ltg = l_t * grad;
    const double fi = STRICT ? -rho / grad : -rho * rcp_tol(grad);   // used only where grad >= TVL1_GRAD_IS_ZERO
    double d1, d2;
    if (rho < -ltg)                     { d1 = l_t * ix;  d2 = l_t * iy; }
    else if (rho > ltg)                 { d1 = -l_t * ix; d2 = -l_t * iy; }
    else if (grad < TVL1_GRAD_IS_ZERO)  { d1 = 0.0;       d2 = 0.0; }
    else                                { d1 = fi * ix;   d2 = fi * iy; }
    const double v1 = u.x + d1, v2 = u.y + d2;
    return make_double2(rnd_to<T>(v1 + theta * div1), rnd_to<T>(v2 + theta * div2));
}

#ifdef OFX_DIV_SHARED
// 1 / d refined exactly as the compiler refines it inside an f64 division (v_rcp_f64 + two Newton steps in FMA)
OFX_DEV double rcp_refined(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
// n / d from the refined reciprocal: quotient estimate, remainder, correction (v_div_fmas without scaling = this FMA),
// v_div_fixup for zeros / infinities / NaNs and the sign
OFX_DEV double div_by_rcp(double n, double d, double r)
{
    double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    q = __builtin_fma(rem, r, q);
    return __builtin_amdgcn_div_fixup(q, d, n);
}
#endif

// Stage "dual" at one pixel: forward gradient of the NEW u (src/operators.cpp:86-125) and the dual
// update (src/tvl1flow.cpp:169-181).  un = new u here, r1/r2 = new u1/u2 of the right pixel, dn = new u
// of the pixel below.
// T = double (strict mode): glibc-exact hypot and four IEEE divisions, bit-identical to the reference.
// T = float  (fast mode): the storage already rounds every result to 24 bits, so the last-bit fidelity of
// the double arithmetic buys nothing; hypot is sqrt(x*x + y*y) and each denominator is inverted once
// (2 divisions instead of 6 per pixel).  This stage is ~60 % of the kernel's VALU work, and the kernel is
// VALU-bound, so the fast mode runs ~1.5x faster; AEPE vs the double reference stays ~1e-5 (tests).
// STRICT = false with T = double is the "tolerance" mode (option "relaxed_dual"): double storage, the fast mode's dual stage.
template <typename T, bool STRICT>
OFX_DEV void tvl1_dual(double2 p1, double2 p2, double2 un, double r1, double r2, double2 dn, bool rig, bool lastrow,
                       double taut, double2 &q1, double2 &q2)
{
#ifdef OFX_CEIL_MEM
    q1 = make_double2(p1.x + 1e-300 * (un.x + r1), p1.y + 1e-300 * dn.x);
    q2 = make_double2(p2.x + 1e-300 * (un.y + r2), p2.y + 1e-300 * dn.y);
    return;
#endif
    const double u1x = rig ? 0.0 : r1 - un.x;
    const double u2x = rig ? 0.0 : r2 - un.y;
    const double u1y = lastrow ? 0.0 : dn.x - un.x;
    const double u2y = lastrow ? 0.0 : dn.y - un.y;
    if (STRICT) {
        const double g1 = hypot_ref(u1x, u1y);
        const double g2 = hypot_ref(u2x, u2y);
        const double ng1 = 1.0 + taut * g1;
        const double ng2 = 1.0 + taut * g2;
#ifdef OFX_DIV_SHARED
        // EXPERIMENT (tools/ab_div_shared.sh; measured and NOT enabled): the four IEEE quotients with the
        // reciprocal refinement shared per denominator.  The sequence is the compiler's own f64 division with its
        // v_div_scale steps removed, which are the identity unless a numerator is tiny / huge or the denominator huge:
        // a wave-uniform guard on the results sends every other case (and nothing else) to the compiler's division.
        const double n1x = p1.x + taut * u1x, n1y = p1.y + taut * u1y, n2x = p2.x + taut * u2x, n2y = p2.y + taut * u2y;
        const double r1 = rcp_refined(ng1), r2 = rcp_refined(ng2);
        q1.x = div_by_rcp(n1x, ng1, r1);
        q1.y = div_by_rcp(n1y, ng1, r1);
        q2.x = div_by_rcp(n2x, ng2, r2);
        q2.y = div_by_rcp(n2y, ng2, r2);
        const double lo = fmin(fmin(fabs(q1.x), fabs(q1.y)), fmin(fabs(q2.x), fabs(q2.y)));
        const double hi = fmax(fmax(fabs(q1.x), fabs(q1.y)), fmax(fabs(q2.x), fabs(q2.y)));
        bool ok = lo >= 0x1p-900 && hi <= 0x1p600 && fmax(ng1, ng2) <= 0x1p50;
        if (!__all(ok)) {                                   // rare: a zero numerator (exact with v_div_fixup), or a real outlier
            ok = fmax(ng1, ng2) <= 0x1p50 && hi <= 0x1p600 && (n1x == 0.0 || fabs(q1.x) >= 0x1p-900) &&
                 (n1y == 0.0 || fabs(q1.y) >= 0x1p-900) && (n2x == 0.0 || fabs(q2.x) >= 0x1p-900) &&
                 (n2y == 0.0 || fabs(q2.y) >= 0x1p-900);
            if (!__all(ok)) {
                q1.x = n1x / ng1;
                q1.y = n1y / ng1;
                q2.x = n2x / ng2;
                q2.y = n2y / ng2;
            }
        }
#else
        q1.x = (p1.x + taut * u1x) / ng1;
        q1.y = (p1.y + taut * u1y) / ng1;
        q2.x = (p2.x + taut * u2x) / ng2;
        q2.y = (p2.y + taut * u2y) / ng2;
#endif
    } else {
        // no bit-exactness to keep here: the square root and the reciprocal are one refinement step on the rsq / rcp estimates
        // (sqrt_tol / rcp_tol, ~2^-45), without range scaling and fix-ups -- the clamp keeps rsq finite at |grad u| = 0, where
        // sqrt(2^-600) = 2^-300 vanishes in 1 + taut g; the denominators are >= 1
        const double g1 = sqrt_tol(fmax(u1x * u1x + u1y * u1y, 0x1p-600));
        const double g2 = sqrt_tol(fmax(u2x * u2x + u2y * u2y, 0x1p-600));
        const double i1 = rcp_tol(1.0 + taut * g1);
        const double i2 = rcp_tol(1.0 + taut * g2);
        q1.x = (p1.x + taut * u1x) * i1;
        q1.y = (p1.y + taut * u1y) * i1;
        q2.x = (p2.x + taut * u2x) * i2;
        q2.y = (p2.y + taut * u2y) * i2;
    }
}

// Stopping test shared by both iteration kernels.  Launch k runs only if NEITHER of the two previous
// iterations ended the loop: with two iterations fused per launch the second one always executes, so a
// loop that ends on the first iteration of a pair leaves a live error in the second slot -- looking
// two slots back keeps every later launch a no-op (and the pair's input buffers intact for the redo).
// e1 = error of iteration k - 1 (0 when k = 0): the fused kernel predicts from it whether the loop may stop on ITS
// first iteration (tvl1_store_a).
OFX_DEV bool tvl1_continues(double prev1, double prev2, int k, int size, double eps2, double &e1)
{
    e1 = 0.0;
    if (k >= 1) {
        e1 = loop_error_from_sum(wave_allreduce_sum(prev1), size, OFX_CRIT_MEAN);
        if (!(e1 > eps2)) return false;
    }
    if (k >= 2 && !(loop_error_from_sum(wave_allreduce_sum(prev2), size, OFX_CRIT_MEAN) > eps2)) return false;
    return true;
}
// The fused kernel always executes both iterations of its pair, so a loop that stops on the FIRST one (odd n) needs
// the intermediate state u_A / p_A, which normally never leaves the registers.  A launch therefore also stores that
// state into the third buffer of the rotation when the stop is plausible: the previous error is within a factor
// `afac` of the threshold (errors decay by 6-14 % per iteration near convergence) or the host asks for it (bit of
// `amask`: first launch of a loop whose predecessor stopped at once).  If the loop then stops on the first
// iteration the host just points the level at that buffer; a miss falls back to recomputing the single iteration.
// Same formula in k_loop_finalize, which tells the host whether the stopping launch had stored its A state.
// afac = 0 switches the mechanism off (fixed-work mode, option store_a = 0): without the guard a loop that has reached its
// bitwise fixed point (e1 == 0) would satisfy 0 <= 0 and store the state in every launch.
OFX_DEV bool tvl1_store_a(int k, double e1, double eps2, double afac) { return afac > 0.0 && k >= 1 && e1 <= eps2 * afac; }

// ---- lockstep groups ------------------------------------------------------------------------------------
// Every array of a level holds G image pairs back to back (pair g at element offset g * nx * ny) and one
// launch serves all of them: blockIdx.y = pair.  The pairs of a group run the same launches k = 0, 1, ...
// but converge on their own: each has its own error slots (err + g * err_stride) and its own position in the
// buffer rotation (the host knows every pair's n at the end of each loop, so it keeps the positions itself).  A
// pair whose loop has ended just returns.
// The three buffers of a state array rotate: launch unit j (a fused pair, or a single iteration) of pair g reads
// buffer (b_g + j) % 3, writes (b_g + j + 1) % 3, and (b_g + j + 2) % 3 -- the output of the unit after next --
// receives the intermediate state when tvl1_store_a says so.  `code` holds the index read in THIS launch, two bits
// per pair.
template <typename V> struct Tri {
    V *b0, *b1, *b2;
};
template <typename V> struct TriSel {
    const V *in;
    V       *out, *alt;
};
template <typename V> OFX_DEV TriSel<V> pick3(const Tri<V> &t, unsigned code, int g, size_t n)
{
    const unsigned i = (code >> (2 * g)) & 3u;
    TriSel<V> r;
    r.in = (i == 0 ? t.b0 : (i == 1 ? t.b1 : t.b2)) + (size_t) g * n;
    r.out = (i == 0 ? t.b1 : (i == 1 ? t.b2 : t.b0)) + (size_t) g * n;
    r.alt = (i == 0 ? t.b2 : (i == 1 ? t.b0 : t.b1)) + (size_t) g * n;
    return r;
}

// ---- one iteration per launch ------------------------------------------------------------------------
// `slot` is where the error of this iteration is accumulated; `check` = index of this iteration in the
// loop (0 = no stopping test, used for the unconditional redo of a single iteration).  Pairs whose bit in
// `runmask` is clear are skipped (the redo runs only on the pairs whose loop ended on an odd count).
#ifdef OFX_ITER1_WAVES
#define OFX_ITER1_ATTR __attribute__((amdgpu_waves_per_eu(OFX_ITER1_WAVES, OFX_ITER1_WAVES)))
#else
#define OFX_ITER1_ATTR
#endif
template <typename T, bool STRICT>
__global__ __launch_bounds__(256) OFX_ITER1_ATTR void k_tvl1_iter(
    Tri<typename Pix<T>::v2> Ut, Tri<typename Pix<T>::v2> P1t, Tri<typename Pix<T>::v2> P2t,
    const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Rg, double *__restrict__ errg, int check, int slot,
    int nx, int ny, int rows, int strips_x, int strips_pad, double l_t, double theta, double taut, double eps2,
    unsigned incode, unsigned runmask, int err_stride)
{
    using v2 = typename Pix<T>::v2;
    const int lane = threadIdx.x & 63;
    const int gw = OFX_WAVE_UNIFORM((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave index
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    const size_t npix = (size_t) nx * ny;
    const TriSel<v2> hu = pick3(Ut, incode, g, npix), h1 = pick3(P1t, incode, g, npix), h2 = pick3(P2t, incode, g, npix);
    const v2 *__restrict__ Uin = hu.in, *__restrict__ P1in = h1.in, *__restrict__ P2in = h2.in;
    v2 *__restrict__ Uout = hu.out, *__restrict__ P1out = h1.out, *__restrict__ P2out = h2.out;
    const v2 *__restrict__ A = Ag + (size_t) g * npix;
    const T *__restrict__ R = Rg + (size_t) g * npix;
    double *__restrict__ err = errg + (size_t) g * err_stride;

    // The previous iterations' error shards are fetched first and tested last, so the ~2 us memory
    // round trip overlaps with the first row's loads instead of preceding them.
    const double prev1 = loop_fetch_prev(err, check), prev2 = loop_fetch_prev(err, check - 1);

    const int strip = gw % strips_pad, band = gw / strips_pad;
    const int y0 = band * rows;
    const bool idle = (strip >= strips_x) || (y0 >= ny);
    const int yend = (y0 + rows < ny) ? y0 + rows : ny;     // rows [y0, yend) are written by this wave

    const int c = strip * STRIP_OUT - 1 + lane;             // lane 0 / 63 are the left / right halo columns
    const int cc = c < 0 ? 0 : (c > nx - 1 ? nx - 1 : c);
    const bool lef = (c == 0), rig = (c == nx - 1);
    const bool owner = (lane >= 1) && (lane <= STRIP_OUT) && (c < nx);

    const unsigned E2 = 2 * sizeof(T);                      // bytes per pixel of the pair arrays
    const unsigned row2 = (unsigned) nx * E2;
    unsigned off = ((unsigned) y0 * nx + cc) * E2;          // byte offset of (y, cc)
    const unsigned offs = ((unsigned) y0 * nx + (c < 0 ? 0 : c)) * E2;   // ... of (y, c) for the stores
    unsigned so = offs;

    // p12 / p22 of the row above the strip (dropped by the top-row rule when y0 == 0)
    double up12 = 0.0, up22 = 0.0;
    RowIn<T> cur;
    if (!idle) {
        if (y0 > 0) {
            up12 = ldw2(at(P1in, off - row2)).y;
            up22 = ldw2(at(P2in, off - row2)).y;
        }
        cur = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off);
    }

    // stopping test of src/tvl1flow.cpp:113
    double e1;
    if (!tvl1_continues(prev1, prev2, check, nx * ny, eps2, e1)) return;
    if (idle) return;

    const unsigned level_bytes = (unsigned) nx * (unsigned) ny * E2;
    const ofx_rsrc rU = make_rsrc(Uout, level_bytes), rP1 = make_rsrc(P1out, level_bytes), rP2 = make_rsrc(P2out, level_bytes);
    double acc = 0.0;
    double2 un_prev = make_double2(0.0, 0.0);                // u_new of row y-1
    double2 p1_prev = make_double2(0.0, 0.0), p2_prev = make_double2(0.0, 0.0);

    for (int y = y0; y <= yend; y++) {
        // prefetch the next row while this one is being processed
        RowIn<T> nxt = cur;
        if (y + 1 <= yend && y + 1 < ny) nxt = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off + row2);

        double2 un = make_double2(0.0, 0.0);
        unsigned st_u = OFX_OOB, st_p = OFX_OOB;
        if (y < ny) {
            const double l11 = wave_shift_up(cur.p1.x);
            const double l21 = wave_shift_up(cur.p2.x);
            un = tvl1_primal<T, STRICT>(cur.u, cur.a, cur.r, cur.p1, cur.p2, l11, l21, up12, up22, lef, rig, y == 0, y == ny - 1,
                                        l_t, theta);
            if (owner && y < yend) {
                st_u = so;
                acc += (un.x - cur.u.x) * (un.x - cur.u.x) + (un.y - cur.u.y) * (un.y - cur.u.y);   // :159-160
            }
        }
        double2 q1 = make_double2(0.0, 0.0), q2 = q1;
        if (y > y0) {
            const double r1 = wave_shift_down(un_prev.x);
            const double r2 = wave_shift_down(un_prev.y);
            tvl1_dual<T, STRICT>(p1_prev, p2_prev, un_prev, r1, r2, un, rig, y - 1 == ny - 1, taut, q1, q2);
            if (owner) st_p = so - row2;
        }
        // always issued; lanes / steps with nothing to write are out of the buffer's range (ofx_device.h)
        bst2<false>(rU, st_u, un, Uout);
        bst2<false>(rP1, st_p, q1, P1out);
        bst2<false>(rP2, st_p, q2, P2out);
        un_prev = un;
        p1_prev = cur.p1;
        p2_prev = cur.p2;
        up12 = cur.p1.y;
        up22 = cur.p2.y;
        cur = nxt;
        off += row2;
        so += row2;
    }
    loop_accumulate(err, slot, acc, gw);
}

// ---- two iterations per launch -----------------------------------------------------------------------
// Iterations k (A) and k+1 (B) fused: the wave marches down its strip once and runs a four-stage
// software pipeline per loaded row y:   S1  u_A(y)   S2  p_A(y-1)   S3  u_B(y-2)   S4  p_B(y-3)
// Intermediate u_A / p_A never touch memory: they live in registers for the one or two marching steps
// until the next stage consumes them.  Each fused iteration widens the dependency cone by one pixel, so
// a wave now owns 60 output columns (lanes 2..61, two halo lanes each side) and loads rows
// y0-1 .. yend+1 (+ p12/p22 of row y0-2).  Halo pixels are recomputed by the neighbouring wave with the
// same arithmetic, so results do not depend on the decomposition and stay bit-identical to the
// one-iteration kernel.  HBM traffic per iteration drops from 15 to ~(9 (rows+3)/rows 64/60 + 6)/2
// elements per pixel (8.3 at rows = 16).
#ifndef STRIP2_OUT
#define STRIP2_OUT 60
#endif
#define STRIP2_HALO ((64 - STRIP2_OUT) / 2)      // halo lanes each side: two are needed
// waves per SIMD the fused kernel is compiled for (3: 130 VGPRs as the compiler likes it; 4: capped at
// 128 VGPRs with a 3-dword spill).  The strip-height model below needs the same number.
#ifndef OFX_ITER2_WAVES
#define OFX_ITER2_WAVES 3
#endif
#if OFX_ITER2_WAVES >= 4
#define OFX_ITER2_BOUNDS __launch_bounds__(256, 4)
#else
#define OFX_ITER2_BOUNDS __launch_bounds__(256)
#endif
// The marching loop of the fused kernel.  SA = true also stores the intermediate state u_A / p_A (tvl1_store_a).
template <typename T, bool NT, bool SA, bool STRICT>
OFX_DEV void tvl1_iter2_march(const typename Pix<T>::v2 *__restrict__ Uin, const typename Pix<T>::v2 *__restrict__ P1in,
                              const typename Pix<T>::v2 *__restrict__ P2in, const typename Pix<T>::v2 *__restrict__ A,
                              const T *__restrict__ R, typename Pix<T>::v2 *Uout, typename Pix<T>::v2 *P1out,
                              typename Pix<T>::v2 *P2out, typename Pix<T>::v2 *Ualt, typename Pix<T>::v2 *P1alt,
                              typename Pix<T>::v2 *P2alt, double *__restrict__ err, int k, int gw, int nx, int ny, int y0,
                              int yend, int ys, int yl, bool lef, bool rig, bool owner, unsigned off, unsigned so,
                              double up12, double up22, RowIn<T> cur, double l_t, double theta, double taut)
{
    const unsigned E2 = 2 * sizeof(T);
    const unsigned row2 = (unsigned) nx * E2;
#ifdef OFX_CEIL_ALU
    const unsigned off0 = off;
#endif
    const double2 z2 = make_double2(0.0, 0.0);
    const unsigned level_bytes = (unsigned) nx * (unsigned) ny * E2;
    const ofx_rsrc rU = make_rsrc(Uout, level_bytes), rP1 = make_rsrc(P1out, level_bytes), rP2 = make_rsrc(P2out, level_bytes);
    const ofx_rsrc rUa = make_rsrc(Ualt, level_bytes), rP1a = make_rsrc(P1alt, level_bytes), rP2a = make_rsrc(P2alt, level_bytes);
    double accA = 0.0, accB = 0.0;
    double2 uA0 = z2, uA1 = z2, uA2 = z2;                    // u_A of rows y, y-1, y-2
    double2 a1 = z2, a2 = z2;                                // (I1wx, I1wy) of rows y-1, y-2
    double r1c = 0.0, r2c = 0.0;                             // rho_c of rows y-1, y-2
    double2 p0a = z2, p0b = z2;                              // p (iteration k-1) of row y-1
    double2 pA1a = z2, pA1b = z2, pA2a = z2, pA2b = z2;      // p_A of rows y-2, y-3
    double2 uB0 = z2, uB1 = z2;                              // u_B of rows y-2, y-3

    for (int y = ys; y <= yend + 2; y++) {
        RowIn<T> nxt = cur;
#ifdef OFX_CEIL_ALU   // ceiling experiment (tools/ceilings.sh): arithmetic kept, loads alternate between the strip's first two rows
        if (y + 1 <= yl) nxt = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off0 + (((unsigned) (y + 1 - ys)) & 1u) * row2);
#else
        if (y + 1 <= yl) nxt = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off + row2);
#endif

        // S1: u_A(y)
        const bool have1 = (y <= yl);
        unsigned sa1 = OFX_OOB, sa2 = OFX_OOB;
        if (have1) {
            const double l11 = wave_shift_up(cur.p1.x);
            const double l21 = wave_shift_up(cur.p2.x);
            uA0 = tvl1_primal<T, STRICT>(cur.u, cur.a, cur.r, cur.p1, cur.p2, l11, l21, up12, up22, lef, rig, y == 0,
                                         y == ny - 1, l_t, theta);
            if (owner && y >= y0 && y < yend) {
                accA += (uA0.x - cur.u.x) * (uA0.x - cur.u.x) + (uA0.y - cur.u.y) * (uA0.y - cur.u.y);
                sa1 = so;
            }
        }
        // S2: p_A(y-1)
        double2 pAna = z2, pAnb = z2;
        if (y - 1 >= ys && y - 1 <= yend && y - 1 <= ny - 1) {
            const double n1 = wave_shift_down(uA1.x);
            const double n2 = wave_shift_down(uA1.y);
            tvl1_dual<T, STRICT>(p0a, p0b, uA1, n1, n2, uA0, rig, y - 1 == ny - 1, taut, pAna, pAnb);
            pAna.x = rnd_to<T>(pAna.x); pAna.y = rnd_to<T>(pAna.y);
            pAnb.x = rnd_to<T>(pAnb.x); pAnb.y = rnd_to<T>(pAnb.y);
            if (owner && y - 1 >= y0 && y - 1 < yend) sa2 = so - row2;
        }
        // S3: u_B(y-2)
        unsigned st3 = OFX_OOB, st4 = OFX_OOB;
        if (y - 2 >= y0 && y - 2 <= yend && y - 2 <= ny - 1) {
            const double l11 = wave_shift_up(pA1a.x);
            const double l21 = wave_shift_up(pA1b.x);
            uB0 = tvl1_primal<T, STRICT>(uA2, a2, r2c, pA1a, pA1b, l11, l21, pA2a.y, pA2b.y, lef, rig, y - 2 == 0,
                                         y - 2 == ny - 1, l_t, theta);
            if (owner && y - 2 < yend) {
                st3 = so - 2 * row2;
                accB += (uB0.x - uA2.x) * (uB0.x - uA2.x) + (uB0.y - uA2.y) * (uB0.y - uA2.y);
            }
        }
        // S4: p_B(y-3)
        double2 q1 = z2, q2 = z2;
        if (y - 3 >= y0 && y - 3 < yend) {
            const double n1 = wave_shift_down(uB1.x);
            const double n2 = wave_shift_down(uB1.y);
            tvl1_dual<T, STRICT>(pA2a, pA2b, uB1, n1, n2, uB0, rig, y - 3 == ny - 1, taut, q1, q2);
            if (owner) st4 = so - 3 * row2;
        }
#ifdef OFX_CEIL_ALU   // ceiling experiment: every store dropped (still issued, out of range), loads stay on the first row
        st3 = OFX_OOB; st4 = OFX_OOB;
#endif
        // the stores of this step: always issued, lanes / steps with nothing to write are out of range
        bst2<NT>(rU, st3, uB0, Uout);
        bst2<NT>(rP1, st4, q1, P1out);
        bst2<NT>(rP2, st4, q2, P2out);
        if (SA) {
            bst2<NT>(rUa, sa1, uA0, Ualt);
            bst2<NT>(rP1a, sa2, pAna, P1alt);
            bst2<NT>(rP2a, sa2, pAnb, P2alt);
        }
        // advance the pipeline by one row
        uA2 = uA1; uA1 = uA0;
        a2 = a1; a1 = cur.a;
        r2c = r1c; r1c = cur.r;
        p0a = cur.p1; p0b = cur.p2;
        up12 = cur.p1.y; up22 = cur.p2.y;
        pA2a = pA1a; pA2b = pA1b; pA1a = pAna; pA1b = pAnb;
        uB1 = uB0;
        cur = nxt;
        off += row2;
        so += row2;
    }
    loop_accumulate(err, k, accA, gw);
    loop_accumulate(err, k + 1, accB, gw);
}

template <typename T, bool NT, bool STRICT>
__global__ OFX_ITER2_BOUNDS void k_tvl1_iter2(
    Tri<typename Pix<T>::v2> Ut, Tri<typename Pix<T>::v2> P1t, Tri<typename Pix<T>::v2> P2t,
    const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Rg, double *__restrict__ errg, int k, int nx, int ny,
    int rows, int strips_x, int strips_pad, double l_t, double theta, double taut, double eps2, unsigned incode,
    unsigned amask, double afac, int err_stride)
{
    using v2 = typename Pix<T>::v2;
    const int lane = threadIdx.x & 63;
    const int gw = OFX_WAVE_UNIFORM((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave index
    const int g = blockIdx.y;
    const size_t npix = (size_t) nx * ny;
    const TriSel<v2> hu = pick3(Ut, incode, g, npix), h1 = pick3(P1t, incode, g, npix), h2 = pick3(P2t, incode, g, npix);
    const v2 *__restrict__ Uin = hu.in, *__restrict__ P1in = h1.in, *__restrict__ P2in = h2.in;
    const v2 *__restrict__ A = Ag + (size_t) g * npix;
    const T *__restrict__ R = Rg + (size_t) g * npix;
    double *__restrict__ err = errg + (size_t) g * err_stride;
    const double prev1 = loop_fetch_prev(err, k), prev2 = loop_fetch_prev(err, k - 1);

    const int band = gw / strips_pad;
#ifdef OFX_XCD_ROT   // A/B knob: rotate the tile columns from band to band, so vertical neighbours sit on different XCDs
    const int strip = (gw % strips_pad + 4 * OFX_XCD_ROT * band) % strips_pad;
#else
    const int strip = gw % strips_pad;
#endif
    const int y0 = band * rows;
    const bool idle = (strip >= strips_x) || (y0 >= ny);
    const int yend = (y0 + rows < ny) ? y0 + rows : ny;     // rows [y0, yend) are written by this wave
    const int ys = y0 > 0 ? y0 - 1 : 0;                      // first row loaded
    const int yl = (yend + 1 < ny - 1) ? yend + 1 : ny - 1;  // last row loaded

    const int c = strip * STRIP2_OUT - STRIP2_HALO + lane;  // lanes 0,1 / 62,63 are halo columns
    const int cc = c < 0 ? 0 : (c > nx - 1 ? nx - 1 : c);
    const bool lef = (c == 0), rig = (c == nx - 1);
    const bool owner = (lane >= STRIP2_HALO) && (lane < STRIP2_OUT + STRIP2_HALO) && (c < nx);

    const unsigned E2 = 2 * sizeof(T);                      // bytes per pixel of the pair arrays
    const unsigned row2 = (unsigned) nx * E2;
    const unsigned off = ((unsigned) ys * nx + cc) * E2;    // byte offset of (y, cc): loads
    const unsigned so = ((unsigned) ys * nx + (c < 0 ? 0 : c)) * E2;   // ... of (y, c): stores (owner lanes only)

    double up12 = 0.0, up22 = 0.0;                           // p12 / p22 (iteration k-1) of row y-1
    RowIn<T> cur;
    if (!idle) {
        if (ys > 0) {
            up12 = ldw2(at(P1in, off - row2)).y;
            up22 = ldw2(at(P2in, off - row2)).y;
        }
        cur = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off);
    }
    double e1;
    if (!tvl1_continues(prev1, prev2, k, nx * ny, eps2, e1)) return;
    if (idle) return;
    // wave-uniform (every wave of the pair reduces the same shards in the same order)
    if (((amask >> g) & 1u) || tvl1_store_a(k, e1, eps2, afac))
        tvl1_iter2_march<T, NT, true, STRICT>(Uin, P1in, P2in, A, R, hu.out, h1.out, h2.out, hu.alt, h1.alt, h2.alt, err, k, gw, nx, ny,
                                      y0, yend, ys, yl, lef, rig, owner, off, so, up12, up22, cur, l_t, theta, taut);
    else
        tvl1_iter2_march<T, NT, false, STRICT>(Uin, P1in, P2in, A, R, hu.out, h1.out, h2.out, hu.alt, h1.alt, h2.alt, err, k, gw, nx, ny,
                                       y0, yend, ys, yl, lef, rig, owner, off, so, up12, up22, cur, l_t, theta, taut);
}

// ---- three iterations per launch ------------------------------------------------------------------------------------------
// The two-iteration kernel in tolerance mode is bound by its memory streams (6.0 - 6.3 TB/s by the counters, VALU 45 % active,
// profiles/r03_pmc_group_launches.json).  A third fused iteration moves the same streams once per THREE iterations: six-stage
// pipeline per loaded row y,   S1 u_A(y)  S2 p_A(y-1)  S3 u_B(y-2)  S4 p_B(y-3)  S5 u_C(y-4)  S6 p_C(y-5),   rows y0-2 .. yend+2
// loaded.  The dependency cone needs three halo lanes each side (58 output columns); the kernel takes FOUR and stores 56 columns:
// 56 x 16 B = seven whole 128-byte lines per row and stream, where 58 columns left a partial line at both ends of every row
// segment -- the launch's memory ceiling (OFX_CEIL_MEM: loads, stores and lane shifts, no arithmetic) 274 -> 249 us at 1080p x 5,
// the launch itself 292 -> 277 us (profiles/r04_ab_iter3_alignment.txt).  HBM elements per pixel and iteration:
// (9 (rows+5)/rows 64/56 + 6) / 3 = 5.9 at 32 rows against 8.4 for two.  CNT < 3 runs only the first CNT iterations (the redo of a
// loop that ended inside a unit and the tail unit of an iteration limit that is no multiple of 3): same stages, results stored
// after stage 2 CNT.
#ifndef STRIP3_OUT
#define STRIP3_OUT 56
#endif
#define STRIP3_HALO ((64 - STRIP3_OUT) / 2)      // halo lanes each side: three are needed
// Round 4 (profiles/r04_ab_iter3_alignment.txt, all on the 1080p x 5 launch): memory ceiling 266 us, ALU ceiling 219 us (256 at
// 2 waves per SIMD -- which rules out a four-iteration kernel of ~205 VGPRs), production 291 us; measured and dropped: loads two
// rows ahead (production + 7 %, ceiling + 4 %), non-temporal row loads (+ 9 %: the halo re-reads then miss), a workgroup barrier
// per step to keep adjacent strips on the same row (+ 1 %), strips of 48 / 64 rows (no change).
// Measured and dropped (profiles/r03_q_ab_three_iterations_per_launch.txt): 2 waves per SIMD instead of 3 (same speed: the
// launch is not bound by resident waves); the pipeline registers in row-indexed rings with a four-step loop body instead of
// shifting ~50 doubles per step (v_mov_b64 119 -> 57 per step, 210 VGPRs: 1.5 % SLOWER -- VALU issue is not the bound either).
#ifndef OFX_ITER3_WAVES
#define OFX_ITER3_WAVES 3
#endif
template <typename T, bool NT, bool STRICT, int CNT>
OFX_DEV void tvl1_iter3_march(const typename Pix<T>::v2 *__restrict__ Uin, const typename Pix<T>::v2 *__restrict__ P1in,
                              const typename Pix<T>::v2 *__restrict__ P2in, const typename Pix<T>::v2 *__restrict__ A,
                              const T *__restrict__ R, typename Pix<T>::v2 *Uout, typename Pix<T>::v2 *P1out,
                              typename Pix<T>::v2 *P2out, double *__restrict__ err, int slot0, int slot_step, int gw, int nx,
                              int ny, int y0, int yend, int ys, int yl, bool lef, bool rig, bool owner, unsigned off, unsigned so,
                              double up12, double up22, RowIn<T> cur, double l_t, double theta, double taut)
{
    const unsigned E2 = 2 * sizeof(T);
    const unsigned row2 = (unsigned) nx * E2;
    const double2 z2 = make_double2(0.0, 0.0);
    const unsigned level_bytes = (unsigned) nx * (unsigned) ny * E2;
    const ofx_rsrc rU = make_rsrc(Uout, level_bytes), rP1 = make_rsrc(P1out, level_bytes), rP2 = make_rsrc(P2out, level_bytes);
    double accA = 0.0, accB = 0.0, accC = 0.0;
    double2 uA0 = z2, uA1 = z2, uA2 = z2;                    // u_A of rows y, y-1, y-2
    double2 a1 = z2, a2 = z2, a3 = z2, a4 = z2;              // (I1wx, I1wy) of rows y-1 .. y-4
    double r1c = 0.0, r2c = 0.0, r3c = 0.0, r4c = 0.0;       // rho_c of rows y-1 .. y-4
    double2 p0a = z2, p0b = z2;                              // p (iteration k-1) of row y-1
    double2 pA1a = z2, pA1b = z2, pA2a = z2, pA2b = z2;      // p_A of rows y-2, y-3
    double2 uB0 = z2, uB1 = z2, uB2 = z2;                    // u_B of rows y-2, y-3, y-4
    double2 pB1a = z2, pB1b = z2, pB2a = z2, pB2b = z2;      // p_B of rows y-4, y-5
    double2 uC0 = z2, uC1 = z2;                              // u_C of rows y-4, y-5
    const int nyl = ny - 1;
    const int yB0 = y0 > 0 ? y0 - 1 : 0;                     // first row of the B stages
    const int ylc = (yend + CNT - 1 < yl) ? yend + CNT - 1 : yl;                    // last row of u_A
    const int hiA = (yend + CNT - 2 < nyl) ? yend + CNT - 2 : nyl;                  // last row of p_A and u_B
    const int hiB = (yend + CNT - 3 < nyl) ? yend + CNT - 3 : nyl;                  // last row of p_B
    const int hiC = (yend < nyl) ? yend : nyl;                                      // last row of u_C
    const int ylast = yend - 1 + (2 * CNT - 1);
#ifdef OFX_CEIL_ALU
    const unsigned off0 = off;
#endif

    auto step = [&](const int y) {
        RowIn<T> nxt = cur;
#ifdef OFX_CEIL_ALU   // ceiling experiment (tools/ceilings.sh): arithmetic kept, loads alternate between the strip's first two rows
        if (y + 1 <= ylc) nxt = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off0 + ((y & 1) ? row2 : 0u));
#else
        if (y + 1 <= ylc) nxt = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off + row2);
#endif
        unsigned stu = OFX_OOB, stp = OFX_OOB;               // this step's stores: u (one row), p (the row above it)
        double2 su = z2, sp1 = z2, sp2 = z2;

        // S1: u_A(y)
        if (y <= ylc) {
            const double l11 = wave_shift_up(cur.p1.x);
            const double l21 = wave_shift_up(cur.p2.x);
            uA0 = tvl1_primal<T, STRICT>(cur.u, cur.a, cur.r, cur.p1, cur.p2, l11, l21, up12, up22, lef, rig, y == 0, y == nyl, l_t,
                                         theta);
            const bool mine = owner && y >= y0 && y < yend;
            const double dA = (uA0.x - cur.u.x) * (uA0.x - cur.u.x) + (uA0.y - cur.u.y) * (uA0.y - cur.u.y);
            accA += mine ? dA : 0.0;
            if (CNT == 1 && mine) { stu = so; su = uA0; }
        }
        // S2: p_A(y-1)
        double2 pAna = z2, pAnb = z2;
        if (y - 1 >= ys && y - 1 <= hiA) {
            const double n1 = wave_shift_down(uA1.x);
            const double n2 = wave_shift_down(uA1.y);
            tvl1_dual<T, STRICT>(p0a, p0b, uA1, n1, n2, uA0, rig, y - 1 == nyl, taut, pAna, pAnb);
            pAna.x = rnd_to<T>(pAna.x); pAna.y = rnd_to<T>(pAna.y);
            pAnb.x = rnd_to<T>(pAnb.x); pAnb.y = rnd_to<T>(pAnb.y);
            if (CNT == 1 && owner && y - 1 >= y0 && y - 1 < yend) { stp = so - row2; sp1 = pAna; sp2 = pAnb; }
        }
        double2 pBna = z2, pBnb = z2;
        if (CNT >= 2) {
            // S3: u_B(y-2)
            if (y - 2 >= yB0 && y - 2 <= hiA) {
                const double l11 = wave_shift_up(pA1a.x);
                const double l21 = wave_shift_up(pA1b.x);
                uB0 = tvl1_primal<T, STRICT>(uA2, a2, r2c, pA1a, pA1b, l11, l21, pA2a.y, pA2b.y, lef, rig, y - 2 == 0, y - 2 == nyl,
                                             l_t, theta);
                const bool mine = owner && y - 2 >= y0 && y - 2 < yend;
                const double dB = (uB0.x - uA2.x) * (uB0.x - uA2.x) + (uB0.y - uA2.y) * (uB0.y - uA2.y);
                accB += mine ? dB : 0.0;
                if (CNT == 2 && mine) { stu = so - 2 * row2; su = uB0; }
            }
            // S4: p_B(y-3)
            if (y - 3 >= yB0 && y - 3 <= hiB) {
                const double n1 = wave_shift_down(uB1.x);
                const double n2 = wave_shift_down(uB1.y);
                tvl1_dual<T, STRICT>(pA2a, pA2b, uB1, n1, n2, uB0, rig, y - 3 == nyl, taut, pBna, pBnb);
                pBna.x = rnd_to<T>(pBna.x); pBna.y = rnd_to<T>(pBna.y);
                pBnb.x = rnd_to<T>(pBnb.x); pBnb.y = rnd_to<T>(pBnb.y);
                if (CNT == 2 && owner && y - 3 >= y0 && y - 3 < yend) { stp = so - 3 * row2; sp1 = pBna; sp2 = pBnb; }
            }
        }
        if (CNT == 3) {
            // S5: u_C(y-4)
            if (y - 4 >= y0 && y - 4 <= hiC) {
                const double l11 = wave_shift_up(pB1a.x);
                const double l21 = wave_shift_up(pB1b.x);
                uC0 = tvl1_primal<T, STRICT>(uB2, a4, r4c, pB1a, pB1b, l11, l21, pB2a.y, pB2b.y, lef, rig, y - 4 == 0, y - 4 == nyl,
                                             l_t, theta);
                const bool mine = owner && y - 4 < yend;
                const double dC = (uC0.x - uB2.x) * (uC0.x - uB2.x) + (uC0.y - uB2.y) * (uC0.y - uB2.y);
                accC += mine ? dC : 0.0;
                stu = mine ? so - 4 * row2 : stu;
                su = uC0;
            }
            // S6: p_C(y-5)
            if (y - 5 >= y0 && y - 5 < yend) {
                const double n1 = wave_shift_down(uC1.x);
                const double n2 = wave_shift_down(uC1.y);
                tvl1_dual<T, STRICT>(pB2a, pB2b, uC1, n1, n2, uC0, rig, y - 5 == nyl, taut, sp1, sp2);
                if (owner) stp = so - 5 * row2;
            }
        }
        // the stores of this step: always issued, lanes / steps with nothing to write are out of range
#ifdef OFX_CEIL_ALU
        stu = OFX_OOB; stp = OFX_OOB;
#endif
        bst2<NT>(rU, stu, su, Uout);
        bst2<NT>(rP1, stp, sp1, P1out);
        bst2<NT>(rP2, stp, sp2, P2out);
        // advance the pipeline by one row
        uA2 = uA1; uA1 = uA0;
        a4 = a3; a3 = a2; a2 = a1; a1 = cur.a;
        r4c = r3c; r3c = r2c; r2c = r1c; r1c = cur.r;
        p0a = cur.p1; p0b = cur.p2;
        up12 = cur.p1.y; up22 = cur.p2.y;
        pA2a = pA1a; pA2b = pA1b; pA1a = pAna; pA1b = pAnb;
        uB2 = uB1; uB1 = uB0;
        pB2a = pB1a; pB2b = pB1b; pB1a = pBna; pB1b = pBnb;
        uC1 = uC0;
        cur = nxt;
        off += row2;
        so += row2;
    };
    for (int y = ys; y <= ylast; y++) step(y);
    loop_accumulate(err, slot0, accA, gw);
    if (CNT >= 2) loop_accumulate(err, slot0 + slot_step, accB, gw);
    if (CNT == 3) loop_accumulate(err, slot0 + 2 * slot_step, accC, gw);
}

// Launch unit of up to three iterations.  Two ways to say which:
//  * dev == nullptr (the re-run of a stopped unit's first iterations): k0, `check`, slot0 and `nit` (iterations to run, 4 bits per
//    pair; all errors into slot0 when check == 0) as arguments -- the interface of k_tvl1_tile;
//  * dev != nullptr, a CURSOR loop (ofx_loop.h): launch number `launch` of the loop; pair g starts at iteration
//    k0 = dev->cursor[launch & 1][g] and runs 3 iterations -- or 2 when the error of iteration k0 - 1 is within afac2 of the
//    threshold, 1 within afac1: the loop is about to stop, and everything computed past the stop is wasted and costs a re-run
//    (the decision is a function of data every wave reads, so all waves of the pair take the same one).  One thread per pair
//    leaves the next launch's cursor and logs where this one started, also when the launch is a no-op.
template <typename T, bool NT, bool STRICT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OFX_ITER3_WAVES, OFX_ITER3_WAVES))) void k_tvl1_iter3(
    Tri<typename Pix<T>::v2> Ut, Tri<typename Pix<T>::v2> P1t, Tri<typename Pix<T>::v2> P2t,
    const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Rg, double *__restrict__ errg, int k0, int check, int slot0,
    int nx, int ny, int rows, int strips_x, int strips_pad, double l_t, double theta, double taut, double eps2, unsigned incode,
    unsigned runmask, unsigned long long nit, int err_stride, OfxLoopDev *dev, int launch, double afac1, double afac2, int max_iter)
{
    using v2 = typename Pix<T>::v2;
    const int lane = threadIdx.x & 63;
    const int gw = OFX_WAVE_UNIFORM((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave index
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    int niter = (int) ((nit >> (4 * g)) & 15ull);
    const bool scribe = dev && blockIdx.x == 0 && threadIdx.x == 0;       // the one thread that writes the pair's loop state
    if (dev) {
        k0 = dev->cursor[launch & 1][g];
        slot0 = k0;
        check = 1;
        if (scribe) {
            dev->ulog[g][launch] = k0;
            dev->cursor[(launch + 1) & 1][g] = k0;                         // unless the launch turns out to do work (below)
        }
        if (k0 >= max_iter) return;
    }
    const size_t npix = (size_t) nx * ny;
    const TriSel<v2> hu = pick3(Ut, incode, g, npix), h1 = pick3(P1t, incode, g, npix), h2 = pick3(P2t, incode, g, npix);
    const v2 *__restrict__ Uin = hu.in, *__restrict__ P1in = h1.in, *__restrict__ P2in = h2.in;
    const v2 *__restrict__ A = Ag + (size_t) g * npix;
    const T *__restrict__ R = Rg + (size_t) g * npix;
    double *__restrict__ err = errg + (size_t) g * err_stride;
    double prev[3];
#pragma unroll
    for (int i = 0; i < 3; i++) prev[i] = (check && k0 - i > 0) ? loop_fetch_prev(err, k0 - i) : 0.0;

    const int band = gw / strips_pad, strip = gw % strips_pad;
    const int y0 = band * rows;
    const bool idle = (strip >= strips_x) || (y0 >= ny);
    const int yend = (y0 + rows < ny) ? y0 + rows : ny;     // rows [y0, yend) are written by this wave
    const int ys = y0 > 2 ? y0 - 2 : 0;                      // first row loaded
    const int yl = (yend + 2 < ny - 1) ? yend + 2 : ny - 1;  // last row loaded
    const int c = strip * STRIP3_OUT - STRIP3_HALO + lane;  // lanes 0..2 / 61..63 are halo columns
    const int cc = c < 0 ? 0 : (c > nx - 1 ? nx - 1 : c);
    const bool lef = (c == 0), rig = (c == nx - 1);
    const bool owner = (lane >= STRIP3_HALO) && (lane < STRIP3_OUT + STRIP3_HALO) && (c < nx);
    const unsigned E2 = 2 * sizeof(T);
    const unsigned row2 = (unsigned) nx * E2;
    const unsigned off = ((unsigned) ys * nx + cc) * E2;    // byte offset of (y, cc): loads
    const unsigned so = ((unsigned) ys * nx + (c < 0 ? 0 : c)) * E2;   // ... of (y, c): stores (owner lanes only)
    double up12 = 0.0, up22 = 0.0;                           // p12 / p22 (iteration k0-1) of row ys-1
    RowIn<T> cur;
    if (!idle) {
        if (ys > 0) {
            up12 = ldw2(at(P1in, off - row2)).y;
            up22 = ldw2(at(P2in, off - row2)).y;
        }
        cur = tvl1_load_row<T>(Uin, P1in, P2in, A, R, off);
    }
    if (check) {                                            // stopping test of src/tvl1flow.cpp:113 -- the same decision in every wave
        double e1 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (k0 - i > 0) {
                const double e = loop_error_from_sum(wave_allreduce_sum(prev[i]), nx * ny, OFX_CRIT_MEAN);
                if (!(e > eps2)) return;
                if (i == 0) e1 = e;
            }
        }
        if (dev) {
            niter = (k0 > 0 && e1 <= eps2 * afac1) ? 1 : ((k0 > 0 && e1 <= eps2 * afac2) ? 2 : 3);
            if (niter > max_iter - k0) niter = max_iter - k0;
            if (scribe) dev->cursor[(launch + 1) & 1][g] = k0 + niter;
        }
    }
    if (idle) return;
    const int step = check ? 1 : 0;                         // a redo's errors all go into its one scratch slot
    if (niter >= 3)
        tvl1_iter3_march<T, NT, STRICT, 3>(Uin, P1in, P2in, A, R, hu.out, h1.out, h2.out, err, slot0, step, gw, nx, ny, y0, yend, ys, yl,
                                           lef, rig, owner, off, so, up12, up22, cur, l_t, theta, taut);
    else if (niter == 2)
        tvl1_iter3_march<T, NT, STRICT, 2>(Uin, P1in, P2in, A, R, hu.out, h1.out, h2.out, err, slot0, step, gw, nx, ny, y0, yend, ys, yl,
                                           lef, rig, owner, off, so, up12, up22, cur, l_t, theta, taut);
    else
        tvl1_iter3_march<T, NT, STRICT, 1>(Uin, P1in, P2in, A, R, hu.out, h1.out, h2.out, err, slot0, step, gw, nx, ny, y0, yend, ys, yl,
                                           lef, rig, owner, off, so, up12, up22, cur, l_t, theta, taut);
}

// ---- K iterations per launch on a 2-D tile: the small pyramid levels ---------------------------------------------------
// On a level of a few thousand pixels a launch of the marching kernels is a latency chain: a strip of r rows takes r + 5
// dependent marching steps of ~2.4 us while most of the chip idles (120x68: 6 us per iteration).  Here the rows are spread
// over the waves of a workgroup instead: a workgroup owns a region of 64 columns x TILE_RH rows, ONE PIXEL PER THREAD (wave =
// row, lane = column), the state of its pixel in registers, horizontal neighbours through wave shifts, vertical neighbours
// through LDS (p12 / p22 to the row below before the primal stage, u_new to the row above before the dual stage: two
// barriers per iteration).  K iterations run back to back; the dependency cone of an iteration is one pixel in every
// direction, so the outer K pixels of the region are a recomputed halo and the workgroup stores the inner (64 - 2 K) x
// (TILE_RH - 2 K) pixels (at the image border the boundary rules cut the cone, nothing more is needed).  Same per-pixel
// functions in the same order as the marching kernels: bit-identical.  The stopping test looks at the K error slots of the
// previous launch; a loop that ends inside a launch is finished by re-running the first n - k0 iterations of that launch
// from its (untouched) input buffers: `nit` holds the iteration count of every pair, 4 bits each.
#define TILE_RH 16
template <typename T, bool STRICT, int K>
__global__ __launch_bounds__(64 * TILE_RH) void k_tvl1_tile(
    Tri<typename Pix<T>::v2> Ut, Tri<typename Pix<T>::v2> P1t, Tri<typename Pix<T>::v2> P2t,
    const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Rg, double *__restrict__ errg, int k0, int check, int slot0,
    int nx, int ny, int tiles_x, double l_t, double theta, double taut, double eps2, unsigned incode, unsigned runmask,
    unsigned long long nit, int err_stride)
{
    using v2 = typename Pix<T>::v2;
    static_assert(K < 16, "4 bits of `nit` per pair");
    __shared__ double2 s_p[2][TILE_RH][64];                  // (p12, p22) of every row, double-buffered over iterations
    __shared__ double2 s_u[2][TILE_RH][64];                  // u_new of every row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    const int niter = (int) ((nit >> (4 * g)) & 15ull);
    const size_t npix = (size_t) nx * ny;
    const TriSel<v2> hu = pick3(Ut, incode, g, npix), h1 = pick3(P1t, incode, g, npix), h2 = pick3(P2t, incode, g, npix);
    double *__restrict__ err = errg + (size_t) g * err_stride;
    // error shards of the previous launch's K iterations (fetched first, tested after the loads are issued)
    double prev[K];
#pragma unroll
    for (int i = 0; i < K; i++) prev[i] = (check && k0 - i > 0) ? loop_fetch_prev(err, k0 - i) : 0.0;

    constexpr int OW = 64 - 2 * K, OH = TILE_RH - 2 * K;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int c = tx * OW - K + lane, y = ty * OH - K + w;
    const int cc = c < 0 ? 0 : (c > nx - 1 ? nx - 1 : c), yc = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    const bool inside = (c >= 0) && (c < nx) && (y >= 0) && (y < ny);
    const bool owner = inside && lane >= K && lane < 64 - K && w >= K && w < TILE_RH - K;
    const bool lef = (c == 0), rig = (c == nx - 1), top = (y == 0), bot = (y == ny - 1);
    const size_t p = (size_t) yc * nx + cc;
    double2 u = ldw2(hu.in + p), p1 = ldw2(h1.in + p), p2 = ldw2(h2.in + p);
    const double2 a = ldw2(Ag + (size_t) g * npix + p);
    const double r = ldw(Rg + (size_t) g * npix + p);
    if (check) {                                            // stopping test of src/tvl1flow.cpp:113 -- the same decision in every wave
#pragma unroll
        for (int i = 0; i < K; i++)
            if (k0 - i > 0 && !(loop_error_from_sum(wave_allreduce_sum(prev[i]), nx * ny, OFX_CRIT_MEAN) > eps2)) return;
    }
    double acc[K];
#pragma unroll
    for (int it = 0; it < K; it++) {
        acc[it] = 0.0;
        if (it < niter) {                                    // niter is uniform over the workgroup: the barriers below are, too
            s_p[it & 1][w][lane] = make_double2(p1.y, p2.y);
            __syncthreads();
            const double2 up = s_p[it & 1][w > 0 ? w - 1 : 0][lane];     // row 0 of the region: halo (or the image's top row)
            const double l11 = wave_shift_up(p1.x), l21 = wave_shift_up(p2.x);
            const double2 un = tvl1_primal<T, STRICT>(u, a, r, p1, p2, l11, l21, up.x, up.y, lef, rig, top, bot, l_t, theta);
            if (owner) acc[it] = (un.x - u.x) * (un.x - u.x) + (un.y - u.y) * (un.y - u.y);   // :159-160
            s_u[it & 1][w][lane] = un;
            __syncthreads();
            const double2 dn = s_u[it & 1][w < TILE_RH - 1 ? w + 1 : w][lane];
            const double r1 = wave_shift_down(un.x), r2 = wave_shift_down(un.y);
            double2 q1, q2;
            tvl1_dual<T, STRICT>(p1, p2, un, r1, r2, dn, rig, bot, taut, q1, q2);
            u = un;
            p1 = make_double2(rnd_to<T>(q1.x), rnd_to<T>(q1.y));
            p2 = make_double2(rnd_to<T>(q2.x), rnd_to<T>(q2.y));
        }
    }
    if (owner) {
        const size_t po = (size_t) y * nx + c;
        stn2(hu.out + po, u);
        stn2(h1.out + po, p1);
        stn2(h2.out + po, p2);
    }
#pragma unroll
    for (int it = 0; it < K; it++)
        if (it < niter) loop_accumulate(err, slot0 + (check ? it : 0), acc[it], blockIdx.x * TILE_RH + w);   // a redo (check = 0): one scratch slot
}

// Warp + linearisation (src/tvl1flow.cpp:94-109): the three bicubic warps of I1, I1x, I1y share one
// set of tap indices and one 4-wide gather per tap; writes A = (I1wx, I1wy) and R = rho_c.
// Thread block of the warp kernel: BX x BY pixels, a wave covers BX x (64 / BX) of them.
#ifndef OFX_WARP_BX
#define OFX_WARP_BX 64
#define OFX_WARP_BY 4
#endif
template <typename T>
__global__ void k_tvl1_warp(const typename Pix<T>::v2 *__restrict__ pag, const T *__restrict__ pbg, const T *__restrict__ I0g,
                            Tri<typename Pix<T>::v2> Ut,
                            typename Pix<T>::v2 *__restrict__ Ag, T *__restrict__ Rg, int nx, int ny, unsigned curcode)
{
    const int j = blockIdx.x * OFX_WARP_BX + threadIdx.x;
    const int i = blockIdx.y * OFX_WARP_BY + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const int g = blockIdx.z;                                // pair of the lockstep group
    const size_t goff = (size_t) g * nx * ny;
    const typename Pix<T>::v2 *__restrict__ pa = pag + goff;
    const T *__restrict__ pb = pbg + goff;
    const T *__restrict__ I0 = I0g + goff;
    const typename Pix<T>::v2 *__restrict__ U = pick3(Ut, curcode, g, (size_t) nx * ny).in;
    typename Pix<T>::v2 *__restrict__ A = Ag + goff;
    T *__restrict__ R = Rg + goff;
    const size_t p = (size_t) i * nx + j;
    const double2 u = ldw2(U + p);
    const BicubicTaps t = bicubic_taps(j + u.x, i + u.y, nx, ny);
    double I1w = 0.0, I1wx = 0.0, I1wy = 0.0;
    if (!t.out) {                                            // border_out = true, :94-96
        bicubic_sample3(pa, pb, t, nx, I1w, I1wx, I1wy);
        I1wx = rnd_to<T>(I1wx);
        I1wy = rnd_to<T>(I1wy);
    }
    stn2(A + p, make_double2(I1wx, I1wy));
    stn(R + p, (I1w - I1wx * u.x - I1wy * u.y - ldw(I0 + p)));       // :107-108
}

// The same warp with the taps staged through LDS.  k_tvl1_warp's time follows the bytes it gathers through the
// vector L1 (16 taps x 24 B = 384 B per pixel, 8.5x the unique data of a block).  Here a block of 32 x 8 pixels first
// finds the bounding box of its taps -- a computed sample has the clean block rows y-1 .. y+2, columns x-1 .. x+2
// (bicubic_sample3) --, loads that box of the (I1, I1x) and I1y planes ONCE with coalesced row loads into LDS and
// gathers from there (LDS: 128 B/clk/CU, consecutive lanes -> consecutive columns, conflict-free).  A block whose box
// does not fit the tile (a motion boundary with a large jump) takes the global-gather path; same arithmetic either
// way, bit-identical results.
// Tile size and register budget are chosen for occupancy: the block has two dependent global-load phases (flow ->
// bounding box -> tile), so its time follows the number of resident waves.  48 x 16 x 24 B = 18 KB lets 8 blocks =
// 32 waves share a CU's LDS, and 64 VGPRs (waves_per_eu 8; 10 spilled) lets them share its registers:
// 1080p 49.0 -> 40.5 us (tile 48 x 24 and 98 VGPRs gave 16 waves per CU; a 16-row block or a 40 x 14 tile: no better).
#ifndef WARP_TW
#define WARP_TW 48          // tile columns
#define WARP_TH 16          // tile rows
#endif
#ifndef WARP_BY
#define WARP_BY 8           // block rows (block = 32 x WARP_BY pixels)
#endif
#ifndef WARP_WAVES
#define WARP_WAVES 8
#endif
#define WARP_ATTR __attribute__((amdgpu_waves_per_eu(WARP_WAVES, WARP_WAVES)))
#define WARP_NT (32 * WARP_BY)
#define WARP_NW (WARP_NT / 64)
template <typename T>
__global__ __launch_bounds__(WARP_NT) WARP_ATTR void k_tvl1_warp_lds(
    const typename Pix<T>::v2 *__restrict__ pag, const T *__restrict__ pbg, const T *__restrict__ I0g,
    Tri<typename Pix<T>::v2> Ut, typename Pix<T>::v2 *__restrict__ Ag, T *__restrict__ Rg, int nx, int ny,
    unsigned curcode)
{
    __shared__ double2 s_a[WARP_TH * WARP_TW];
    __shared__ double s_b[WARP_TH * WARP_TW];
    __shared__ int s_box[WARP_NW][4];                         // per wave: xmin, xmax, ymin, ymax
    const int tid = threadIdx.y * 32 + threadIdx.x, wave = tid >> 6;
    const int j = blockIdx.x * 32 + threadIdx.x;
    const int i = blockIdx.y * WARP_BY + threadIdx.y;
    const bool inside = (j < nx) && (i < ny);
    const int g = blockIdx.z;
    const size_t goff = (size_t) g * nx * ny;
    const typename Pix<T>::v2 *__restrict__ pa = pag + goff;
    const T *__restrict__ pb = pbg + goff;
    const typename Pix<T>::v2 *__restrict__ U = pick3(Ut, curcode, g, (size_t) nx * ny).in;
    const size_t p = (size_t) (inside ? i : 0) * nx + (inside ? j : 0);
    double2 u = make_double2(0.0, 0.0);
    BicubicTaps t;
    t.out = true;
    if (inside) {
        u = ldw2(U + p);
        t = bicubic_taps(j + u.x, i + u.y, nx, ny);
    }
    const bool need = inside && !t.out;
    // bounding box of the taps of the block
    int xlo = need ? t.col[0] : 0x7fffffff, xhi = need ? t.col[3] : -1;
    int ylo = need ? t.row[0] : 0x7fffffff, yhi = need ? t.row[3] : -1;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        xlo = min(xlo, __shfl_xor(xlo, m, 64)); xhi = max(xhi, __shfl_xor(xhi, m, 64));
        ylo = min(ylo, __shfl_xor(ylo, m, 64)); yhi = max(yhi, __shfl_xor(yhi, m, 64));
    }
    if ((tid & 63) == 0) { s_box[wave][0] = xlo; s_box[wave][1] = xhi; s_box[wave][2] = ylo; s_box[wave][3] = yhi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WARP_NW; w++) {
        xlo = min(xlo, s_box[w][0]); xhi = max(xhi, s_box[w][1]);
        ylo = min(ylo, s_box[w][2]); yhi = max(yhi, s_box[w][3]);
    }
    const int W = xhi - xlo + 1, H = yhi - ylo + 1;
    const bool any = (xhi >= 0);
    const bool tile = any && W <= WARP_TW && H <= WARP_TH;    // block-uniform
    if (tile) {
        for (int k = tid; k < W * H; k += WARP_NT) {
            const int r = k / W, c = k - r * W;
            const size_t src = (size_t) (ylo + r) * nx + (xlo + c);
            s_a[r * WARP_TW + c] = ldw2(pa + src);
            s_b[r * WARP_TW + c] = ldw(pb + src);
        }
        __syncthreads();
    }
    if (!inside) return;
    double I1w = 0.0, I1wx = 0.0, I1wy = 0.0;
    if (need) {
        if (tile) {
            const int cx = t.col[0] - xlo, cy = t.row[0] - ylo;
            double c0[4], c1[4], c2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int o = cy * WARP_TW + cx + k;
                const double2 a0 = s_a[o], a1 = s_a[o + WARP_TW], a2 = s_a[o + 2 * WARP_TW], a3 = s_a[o + 3 * WARP_TW];
                const double b0 = s_b[o], b1 = s_b[o + WARP_TW], b2 = s_b[o + 2 * WARP_TW], b3 = s_b[o + 3 * WARP_TW];
                c0[k] = cubic_cell(a0.x, a1.x, a2.x, a3.x, t.fy);
                c1[k] = cubic_cell(a0.y, a1.y, a2.y, a3.y, t.fy);
                c2[k] = cubic_cell(b0, b1, b2, b3, t.fy);
            }
            I1w = cubic_cell(c0[0], c0[1], c0[2], c0[3], t.fx);
            I1wx = cubic_cell(c1[0], c1[1], c1[2], c1[3], t.fx);
            I1wy = cubic_cell(c2[0], c2[1], c2[2], c2[3], t.fx);
        } else {
            bicubic_sample3(pa, pb, t, nx, I1w, I1wx, I1wy);
        }
        I1wx = rnd_to<T>(I1wx);
        I1wy = rnd_to<T>(I1wy);
    }
    stn2(Ag + goff + p, make_double2(I1wx, I1wy));
    stn(Rg + goff + p, (I1w - I1wx * u.x - I1wy * u.y - ldw(I0g + goff + p)));       // :107-108
}

// zoom_in of the flows of a whole group (src/zoom.cpp:132-155 + the `*= 1 / zfactor` of src/tvl1flow.cpp:302-309; the per-
// pixel arithmetic of k_zoom_in_flow, ofx_ops.hip): pair g reads its live buffer of the coarse level's rotation and writes
// buffer 0 of the fine level
template <typename T>
__global__ void k_tvl1_zoom_in_g(Tri<typename Pix<T>::v2> Ut, unsigned curcode, typename Pix<T>::v2 *__restrict__ Uout, int nx,
                                 int ny, int nxx, int nyy, double fx, double fy, double scale)
{
    const int j1 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    if (j1 >= nxx || i1 >= nyy) return;
    const int g = blockIdx.z;
    const typename Pix<T>::v2 *__restrict__ U = pick3(Ut, curcode, g, (size_t) nx * ny).in;
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
    r.x = cubic_cell(c1[0], c1[1], c1[2], c1[3], t.fx) * scale;
    r.y = cubic_cell(c2[0], c2[1], c2[2], c2[3], t.fx) * scale;
    stn2(Uout + (size_t) g * nxx * nyy + (size_t) i1 * nxx + j1, r);
}

// the .flo payloads of a whole group: (u, v) -> float32 pairs (the cast of src/tvl1flow_main.cpp:209-213), pair g from its
// live buffer into the caller's g-th output array
template <typename T>
__global__ void k_tvl1_to_flo_g(Tri<typename Pix<T>::v2> Ut, unsigned curcode, OfxGroupPtrs out, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = blockIdx.y;
    const double2 v = ldw2(pick3(Ut, curcode, g, n).in + i);
    ((float2 *) out.a[g])[i] = make_float2((float) v.x, (float) v.y);
}

// ---- host side ---------------------------------------------------------------------------------------
// One level of a lockstep group: every array holds G pairs back to back (pair g at element g * nx * ny).
template <typename T> struct Tvl1Level {
    using v2 = typename Pix<T>::v2;
    using v4 = typename Pix<T>::v4;
    int nx, ny, G;
    T  *I0, *I1;
    v2 *pa;         // (I1, I1x): gathered by the warp
    T  *pb;         // I1y
    v2 *U[3], *P1[3], *P2[3], *A;
    T  *R;
    unsigned cur;   // two bits per pair: which buffer of the rotation holds the live u / p of pair g
    int last_n[OFX_MAX_GROUP];   // iterations of the previous loop of pair g at this level (0 = none yet)
    size_t n() const { return (size_t) nx * ny; }
    unsigned curidx(int g) const { return (cur >> (2 * g)) & 3u; }
    v2 *Ucur(int g) const { return U[curidx(g)] + (size_t) g * n(); }
    v2 *P1cur(int g) const { return P1[curidx(g)] + (size_t) g * n(); }
    v2 *P2cur(int g) const { return P2[curidx(g)] + (size_t) g * n(); }
    Tri<v2> Ut() const { return Tri<v2>{U[0], U[1], U[2]}; }
    Tri<v2> P1t() const { return Tri<v2>{P1[0], P1[1], P1[2]}; }
    Tri<v2> P2t() const { return Tri<v2>{P2[0], P2[1], P2[2]}; }
};

struct Tvl1Params {
    double tau, lambda, theta, epsilon;
    int warps, verbose, max_iter;
    bool fixed;     // run exactly max_iter iterations (stopping test disabled)
};

template <typename T> static int tvl1_level_alloc(ofx_ctx *ctx, Tvl1Level<T> &L, int nx, int ny, int G, bool images)
{
    const size_t n = (size_t) nx * ny * G;
    L.nx = nx;
    L.ny = ny;
    L.G = G;
    L.cur = 0;
    for (int g = 0; g < OFX_MAX_GROUP; g++) L.last_n[g] = 0;
    L.I0 = L.I1 = nullptr;
    if (images) {
        OFX_TRY(ofx_alloc(ctx, n, &L.I0));
        OFX_TRY(ofx_alloc(ctx, n, &L.I1));
    }
    OFX_TRY(ofx_alloc(ctx, n, &L.pa));
    OFX_TRY(ofx_alloc(ctx, n, &L.pb));
    for (int h = 0; h < 3; h++) {
        OFX_TRY(ofx_alloc(ctx, n, &L.U[h]));
        OFX_TRY(ofx_alloc(ctx, n, &L.P1[h]));
        OFX_TRY(ofx_alloc(ctx, n, &L.P2[h]));
    }
    OFX_TRY(ofx_alloc(ctx, n, &L.A));
    OFX_TRY(ofx_alloc(ctx, n, &L.R));
    return OFX_OK;
}

static int tvl1_pick_rows(const ofx_ctx *ctx, int nx, int ny, int G)
{
    if (ctx->rows_per_wave > 0) return ctx->rows_per_wave;
    // Measured on MI355X (tools/tune_iter.py, profiles/r01_b_rows_sweep.txt): short strips win.  Many
    // short-lived waves hide memory latency better than few long ones (each wave keeps only one row of
    // prefetch in flight), and on the small pyramid levels the launch is a pure latency chain of
    // (rows + 1) dependent marching steps, so the strip height must shrink with the image.
    const long strips = (long) ofx_cdiv(nx, STRIP_OUT) * G;
    if (strips * ofx_cdiv(ny, 4) >= 2048) return 4;
    if (strips * ofx_cdiv(ny, 2) >= 1024) return 2;
    return 1;
}

// Strip height of the fused two-iteration kernel.  A wave of a strip of r rows executes r + 3 marching
// steps (+ ~2 steps of start-up).  At 142 VGPRs the chip holds 3 waves per SIMD = 3072 waves at once, so a
// launch of W waves takes k = ceil(W / 3072) rounds of about (r + 5) step times.  The best strip height is
// therefore the SMALLEST r whose wave count still fits k rounds exactly (1080p: r = 12 -> 32 x 90 = 2880
// waves, one round; r = 8 -> 4320 waves = 1.4 rounds is 9 % slower), minimised over k.  Measured on the
// bench workload at every pyramid level: profiles/r01_g_rows2_on_bench_workload.txt.  A lockstep group of
// G pairs is G times the waves of one pair.
static int tvl1_pick_rows2(const ofx_ctx *ctx, int nx, int ny, int G)
{
    if (ctx->rows_per_wave2 > 0) return ctx->rows_per_wave2;
    const long strips_pad = (long) ofx_cdiv(ofx_cdiv(nx, STRIP2_OUT), 4) * 4 * G;
    const int rmax = 24;
    int best = rmax;
    long best_cost = -1;
    if (ctx->concurrency > 1) {
        // Several contexts share the device (option "concurrency"): other pairs' waves fill this launch's
        // partial rounds, so what counts is the total work  ~ ceil(W / 1024 SIMDs) * (r + 5)  -- taller
        // strips, less halo recomputation (measured with 4 pairs in flight: +8 % job throughput over the
        // single-pair choice).
        static const int cand[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 20, 24};
        for (int r : cand) {
            const long waves = strips_pad * ofx_cdiv(ny, r);
            const long slots = ctx->rows_slots > 0 ? ctx->rows_slots : 1024;
            const long cost = ((waves + slots - 1) / slots) * (r + 5);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
        }
        return best;
    }
    const long slots = 1024L * OFX_ITER2_WAVES;
    for (int k = 1; k <= 4; k++) {
        for (int r = 1; r <= rmax; r++) {
            if (strips_pad * ofx_cdiv(ny, r) > k * slots) continue;
            const long cost = (long) k * (r + 5);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
            break;
        }
    }
    // very short strips triple the halo work; keep 3 rows as long as that still leaves >= 512 waves
    if (best < 3 && strips_pad * ofx_cdiv(ny, 3) >= 512) best = 3;
    return best;
}

static int tvl1_pick_chunk(const ofx_ctx *ctx, int nx, int ny, int G)
{
    if (ctx->chunk > 0) return ctx->chunk;
    // one poll (event wait + host wake-up) costs 20-40 us: a chunk should take about that long, because
    // every launch behind the iteration that ends the loop is a wasted ~2.5 us no-op
    const double est_us = fmax(5.0, (double) nx * ny * G * 120.0 / 4.0e6);   // ~4 TB/s, small-level floor 5 us
    int c = (int) (40.0 / est_us);
    return c < 4 ? 4 : (c > 50 ? 50 : c);
}

// Levels the tile kernel serves: option "tile" = 0 never, K = 4 | 6 fused iterations when the group's level has at most
// "tile_max_px" pixels (default: 4 iterations up to 200 000 pixels x pairs -- 480x270 alone, 240x135 in a group of 5).
static int tvl1_pick_tile(const ofx_ctx *ctx, int nx, int ny, int G)
{
    if (!ctx->tile || !ctx->fuse2) return 0;
    const double lim = ctx->tile_max_px > 0 ? ctx->tile_max_px : 200000.0;
    if ((double) nx * ny * G > lim) return 0;
    return ctx->tile == 6 ? 6 : 4;
}

template <typename T> struct Tvl1Level;
struct Tvl1Params;
template <typename T>
static int tvl1_run_iterations_tile(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, LoopSpec S, int K, int *n_out,
                                    double *err_out, float *ms_out, int *alt_out);
template <typename T>
static int tvl1_run_iterations_tri(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, LoopSpec S, int *n_out, double *err_out,
                                   float *ms_out, int *alt_out);

// strip height of the three-iteration kernel: tvl1_pick_rows2's model with r + 7 marching steps per strip
static int tvl1_pick_rows3(const ofx_ctx *ctx, int nx, int ny, int G)
{
    if (ctx->rows_per_wave3 > 0) return ctx->rows_per_wave3;
    {   // A/B knob only: OFX_ROWS3="ny:rows/ny:rows" forces the strip height on the levels of the given heights
        static const char *tab = getenv("OFX_ROWS3");
        for (const char *q = tab; q && *q;) {
            int lny = 0, r = 0;
            if (sscanf(q, "%d:%d", &lny, &r) == 2 && lny == ny && r > 0) return r;
            q = strchr(q, '/');
            if (q) q++;
        }
    }
    const long strips_pad = (long) ofx_cdiv(ofx_cdiv(nx, STRIP3_OUT), 4) * 4 * G;
    const int rmax = 32;
    int best = rmax;
    long best_cost = -1;
    if (ctx->concurrency > 1) {
        static const int cand[] = {2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40, 48, 54, 64, 72, 80, 96};
        for (int r : cand) {
            if (r > ctx->rows3_max) break;
            const long waves = strips_pad * ofx_cdiv(ny, r);
            const long slots = ctx->rows_slots > 0 ? ctx->rows_slots : 1024;
            const long cost = ((waves + slots - 1) / slots) * (r + 7);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
        }
        return best;
    }
    const long slots = 1024L * OFX_ITER3_WAVES;
    for (int k = 1; k <= 4; k++) {
        for (int r = 1; r <= rmax; r++) {
            if (strips_pad * ofx_cdiv(ny, r) > k * slots) continue;
            const long cost = (long) k * (r + 7);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
            break;
        }
    }
    if (best < 4 && strips_pad * ofx_cdiv(ny, 4) >= 512) best = 4;
    return best;
}

// The inner loop of one warp (src/tvl1flow.cpp:111-182) for all pairs of the group in lockstep.  On return
// L.cur points at the halves holding the results; n_out[g] / err_out[g] are what the reference prints.
template <typename T>
static int tvl1_run_iterations(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, int *n_out, double *err_out,
                               float *ms_out, int *alt_out = nullptr,     // alt_out[g]: 0 stop at the end of a launch unit, 1 inside + recomputed, 2 inside + stored
                               int *unit_out = nullptr)                   // iterations per launch unit of the kernel that ran
{
    const int nx = L.nx, ny = L.ny, G = L.G;
    if ((long long) nx * ny >= (1LL << 27)) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: image larger than 2^27 pixels");
    const double l_t = P.lambda * P.theta, taut = P.tau / P.theta, theta = P.theta;
    const bool pairs = ctx->fuse2 != 0;
    // geometry of the one-iteration kernel (also used for a trailing single iteration and the redo)
    const int rows = tvl1_pick_rows(ctx, nx, ny, G);
    const int strips_x = ofx_cdiv(nx, STRIP_OUT), strips_pad = ofx_cdiv(strips_x, 4) * 4;
    const unsigned gx1 = (unsigned) (strips_pad / 4) * ofx_cdiv(ny, rows);
    const dim3 block(256);
    // geometry of the fused two-iteration kernel
    const int rows2 = tvl1_pick_rows2(ctx, nx, ny, G);
    const int strips2_x = ofx_cdiv(nx, STRIP2_OUT), strips2_pad = ofx_cdiv(strips2_x, 4) * 4;
    const dim3 grid2((unsigned) (strips2_pad / 4) * ofx_cdiv(ny, rows2), G);
    LoopSpec S;
    S.max_iter = P.max_iter;
    S.size = nx * ny;
    S.thr = P.epsilon * P.epsilon;
    S.crit = OFX_CRIT_MEAN;
    S.chunk = tvl1_pick_chunk(ctx, nx, ny, G);
    S.fixed = P.fixed;
    S.pairs = pairs;
    // small levels: K iterations per launch on 2-D tiles (k_tvl1_tile) instead of the marching strips
    const int tileK = tvl1_pick_tile(ctx, nx, ny, G);
    if (unit_out) *unit_out = tileK ? tileK : (pairs ? 2 : 1);
    if (tileK) return tvl1_run_iterations_tile<T>(ctx, L, P, S, tileK, n_out, err_out, ms_out, alt_out);
    // Three iterations per launch (k_tvl1_iter3; option "fuse3": 0 never, 1 on every level of at least fuse3_min_px pixels x
    // pairs, 2 = by measurement, the default).  Measured on one box, interleaved (profiles/r03_q_ab_three_iterations_per_launch.txt):
    // the tolerance mode's job 55.8k -> 61.8k Mpix*warp-iters/s, fixed work 79.8k -> 103.7k; the strict mode, bound by FP64 issue
    // and not by its streams, LOSES 14 % (more halo arithmetic), and a lone pair gains nothing (its launches are latency chains
    // of r + 7 instead of r + 5 marching steps): so 2 = not strict, pairs in lockstep or contexts sharing the device, and a level
    // of at least 500 000 pixels x pairs.
    {
        const bool strict_mode = sizeof(T) == sizeof(double) && !ctx->relaxed_dual;
        const double px = (double) nx * ny * G;
        const bool tri = ctx->fuse3 == 1 ? px >= ctx->fuse3_min_px
                                         : (ctx->fuse3 == 2 && !strict_mode && (G >= 2 || ctx->concurrency > 1) &&
                                            px >= (ctx->fuse3_min_px > 0 ? ctx->fuse3_min_px : 5e5));
        if (pairs && tri) {
            if (unit_out) *unit_out = 3;
            return tvl1_run_iterations_tri<T>(ctx, L, P, S, n_out, err_out, ms_out, alt_out);
        }
    }
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    // one launch touches 15 storage elements per pixel; beyond the Infinity Cache its output is streamed out
    const bool nt_stores = ctx->nt_stores ? ctx->nt_stores == 1 : (double) nx * ny * G * 15.0 * sizeof(T) > 300e6;
    // Launch unit j (a fused pair of iterations, or a single one) of pair g reads buffer (b_g + j) % 3 of the rotation
    // and writes (b_g + j + 1) % 3; (b_g + j + 2) % 3 receives the intermediate state of a fused pair (tvl1_store_a).
    unsigned b0[OFX_MAX_GROUP];
    for (int g = 0; g < G; g++) b0[g] = L.curidx(g);
    auto code_of = [&](unsigned unit) -> unsigned {
        unsigned c = 0;
        for (int g = 0; g < G; g++) c |= ((b0[g] + unit) % 3u) << (2 * g);
        return c;
    };
    // intermediate state: stored when the previous error is within this factor of the threshold ...
    // (option "store_a": 0 = never, 2 = always -- 1e300 stands for "any error" -- for the tests)
    const bool sa = pairs && !P.fixed && ctx->store_a != 0;
    S.afac = sa ? (ctx->store_a == 2 ? 1e300 : 1.5) : 0.0;
    // ... and by the first launch of a loop whose predecessor (previous warp, same level) stopped within two iterations
    unsigned amask0 = 0;
    if (sa)
        for (int g = 0; g < G; g++)
            if (ctx->store_a == 2 || (L.last_n[g] >= 1 && L.last_n[g] <= 2)) amask0 |= 1u << g;
    // strict: the reference's dual update bit for bit (double storage, option "relaxed_dual" off)
    const bool strict = sizeof(T) == sizeof(double) && !ctx->relaxed_dual;
    auto single = [&](unsigned incode, unsigned runmask, int check, int slot, double thr) -> int {
        auto kern = k_tvl1_iter<T, false>;
        if constexpr (sizeof(T) == sizeof(double)) { if (strict) kern = k_tvl1_iter<T, true>; }
        hipLaunchKernelGGL(kern, dim3(gx1, G), block, 0, ctx->stream, L.Ut(), L.P1t(), L.P2t(), L.A, (const T *) L.R,
                           ctx->d_err, check, slot, nx, ny, rows, strips_x, strips_pad, l_t, theta, taut, thr, incode, runmask,
                           err_stride);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    auto launch = [&](int k, int cnt, double thr) -> int {
        const unsigned incode = code_of((unsigned) (pairs ? k / 2 : k));
        if (cnt == 1) return single(incode, all, k, k, thr);
        const unsigned amask = k == 0 ? amask0 : 0u;
        auto kern = nt_stores ? k_tvl1_iter2<T, true, false> : k_tvl1_iter2<T, false, false>;
        if constexpr (sizeof(T) == sizeof(double)) {
            if (strict) kern = nt_stores ? k_tvl1_iter2<T, true, true> : k_tvl1_iter2<T, false, true>;
        }
        hipLaunchKernelGGL(kern, grid2, block, 0, ctx->stream, L.Ut(), L.P1t(), L.P2t(), L.A, (const T *) L.R, ctx->d_err, k, nx,
                           ny, rows2, strips2_x, strips2_pad, l_t, theta, taut, thr, incode, amask, S.afac, err_stride);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    // pairs whose loop ended on the first iteration of a fused pair WITHOUT its intermediate state in the third
    // buffer: recompute that iteration alone (no stopping test, error into the scratch slot) from the unit's input
    // buffer into its output buffer -- one launch for all of them (runmask)
    auto redo = [&](const int *k_of) -> int {
        unsigned incode = 0, runmask = 0;
        for (int g = 0; g < G; g++) {
            if (k_of[g] < 0) continue;
            runmask |= 1u << g;
            incode |= ((b0[g] + (unsigned) (k_of[g] / 2)) % 3u) << (2 * g);
        }
        return single(incode, runmask, 0, S.max_iter, -1.0);
    };
    int took_alt[OFX_MAX_GROUP];
    OFX_TRY(ofx_run_loop_group(ctx, S, G, launch, redo, n_out, err_out, ms_out, amask0, took_alt));
    unsigned cur = 0;
    for (int g = 0; g < G; g++) {
        const unsigned units = (unsigned) (pairs ? (n_out[g] + 1) / 2 : n_out[g]);
        // normally the result is the output of the last unit = the input of the next; a loop that stopped on the first
        // iteration of a pair whose intermediate state was stored continues from the unit's third buffer instead
        cur |= ((b0[g] + units + (took_alt[g] ? 1u : 0u)) % 3u) << (2 * g);
        L.last_n[g] = n_out[g];
        if (alt_out) alt_out[g] = (pairs && (n_out[g] & 1) && n_out[g] != S.max_iter) ? (took_alt[g] ? 2 : 1) : 0;
    }
    L.cur = cur;
    return OFX_OK;
}

static_assert(OFX_MAX_GROUP * 4 <= 64 && OFX_MAX_GROUP * 2 <= 32,
              "k_tvl1_tile / k_tvl1_iter3 pack 4 bits of iteration count (`nit`, 64 bits) and 2 bits of buffer index (`incode`, 32 bits) per pair");
// tvl1_run_iterations through k_tvl1_tile: launch unit j = iterations [j K, j K + K) reads buffer (b_g + j) % 3 and writes
// (b_g + j + 1) % 3; a loop that ends inside a unit is finished by one more launch that re-runs the unit's first n - j K
// iterations from its input (every pair with its own count).
template <typename T>
static int tvl1_run_iterations_tile(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, LoopSpec S, int K, int *n_out,
                                    double *err_out, float *ms_out, int *alt_out)
{
    const int nx = L.nx, ny = L.ny, G = L.G;
    const double l_t = P.lambda * P.theta, taut = P.tau / P.theta, theta = P.theta;
    const bool strict = sizeof(T) == sizeof(double) && !ctx->relaxed_dual;
    const int OW = 64 - 2 * K, OH = TILE_RH - 2 * K;
    const int tiles_x = ofx_cdiv(nx, OW), tiles_y = ofx_cdiv(ny, OH);
    const dim3 grid((unsigned) (tiles_x * tiles_y), G), block(64 * TILE_RH);
    S.pairs = false;
    S.fuse = K;
    S.afac = 0.0;
    // a launch of K iterations takes ~2 us + K x 1.5 us whatever the level's size; one poll per ~40 us
    if (ctx->chunk <= 0) S.chunk = 6 * K;
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    unsigned b0[OFX_MAX_GROUP];
    for (int g = 0; g < G; g++) b0[g] = L.curidx(g);
    auto go = [&](int k0, int check, int slot0, double thr, unsigned incode, unsigned runmask, unsigned long long nit) -> int {
        auto kern = K == 6 ? k_tvl1_tile<T, false, 6> : k_tvl1_tile<T, false, 4>;
        if constexpr (sizeof(T) == sizeof(double)) {
            if (strict) kern = K == 6 ? k_tvl1_tile<T, true, 6> : k_tvl1_tile<T, true, 4>;
        }
        hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, L.Ut(), L.P1t(), L.P2t(), L.A, (const T *) L.R, ctx->d_err, k0, check,
                           slot0, nx, ny, tiles_x, l_t, theta, taut, thr, incode, runmask, nit, err_stride);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    auto launch = [&](int k, int cnt, double thr) -> int {
        unsigned incode = 0;
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) {
            incode |= ((b0[g] + (unsigned) (k / K)) % 3u) << (2 * g);
            nit |= (unsigned long long) cnt << (4 * g);
        }
        return go(k, 1, k, thr, incode, all, nit);
    };
    auto redo = [&](const int *k_of) -> int {
        unsigned incode = 0, runmask = 0;
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) {
            if (k_of[g] < 0) continue;
            const int n = k_of[g] + 1, j = (n - 1) / K;
            runmask |= 1u << g;
            incode |= ((b0[g] + (unsigned) j) % 3u) << (2 * g);
            nit |= (unsigned long long) (n - j * K) << (4 * g);
        }
        // no stopping test; the errors go to the scratch slot behind the loop's own (all K of them into it: never read)
        return go(0, 0, S.max_iter, -1.0, incode, runmask, nit);
    };
    int took_alt[OFX_MAX_GROUP];
    OFX_TRY(ofx_run_loop_group(ctx, S, G, launch, redo, n_out, err_out, ms_out, 0u, took_alt));
    unsigned cur = 0;
    for (int g = 0; g < G; g++) {
        const unsigned units = (unsigned) ((n_out[g] + K - 1) / K);
        cur |= ((b0[g] + units) % 3u) << (2 * g);
        L.last_n[g] = n_out[g];
        if (alt_out) alt_out[g] = 0;
    }
    L.cur = cur;
    return OFX_OK;
}

// tvl1_run_iterations through k_tvl1_iter3 as a CURSOR loop (ofx_loop.h): launch L reads buffer (b_g + L) % 3 of the rotation and
// writes (b_g + L + 1) % 3 and runs three iterations -- two, or one, once the loop is about to stop (k_tvl1_iter3) --, so where a
// unit starts is device state.  A loop that still ends inside a unit is finished by one more launch that re-runs the unit's first
// n - k0 iterations from its input (every pair with its own count).  Option "fuse3_cursor" = 0: the round-3 scheme, fixed units of
// three iterations with the iteration index passed from the host.
template <typename T>
static int tvl1_run_iterations_tri(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, LoopSpec S, int *n_out, double *err_out,
                                   float *ms_out, int *alt_out)
{
    const int nx = L.nx, ny = L.ny, G = L.G;
    const double l_t = P.lambda * P.theta, taut = P.tau / P.theta, theta = P.theta;
    const bool strict = sizeof(T) == sizeof(double) && !ctx->relaxed_dual;
    const int rows = tvl1_pick_rows3(ctx, nx, ny, G);
    const int strips_x = ofx_cdiv(nx, STRIP3_OUT), strips_pad = ofx_cdiv(strips_x, 4) * 4;
    const dim3 grid((unsigned) (strips_pad / 4) * ofx_cdiv(ny, rows), G), block(256);
    const bool nt_stores = ctx->nt_stores ? ctx->nt_stores == 1 : (double) nx * ny * G * 15.0 * sizeof(T) > 300e6;
    S.pairs = false;
    S.fuse = 3;
    S.afac = 0.0;
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    unsigned b0[OFX_MAX_GROUP];
    for (int g = 0; g < G; g++) b0[g] = L.curidx(g);
    const bool cursor = ctx->fuse3_cursor != 0 && !S.fixed;
    // the error decays by 6-14 % per iteration near the threshold: within afac2 the loop has a handful of iterations left,
    // within afac1 the next one is the last more often than not (A/B on one box, profiles/r04_ab_cursor_loop.txt: fixed units of three 68.1k, cursor loop 71.2-71.6k Mpix*warp-iters/s whatever the two ratios in 1.08-1.2 / 1.3-2.0; 1.2 leaves the fewest re-runs)
    const double afac2 = ctx->fuse3_afac2 > 0 ? ctx->fuse3_afac2 : 1.5, afac1 = ctx->fuse3_afac1 > 0 ? ctx->fuse3_afac1 : 1.2;
    auto go = [&](int k0, int check, int slot0, double thr, unsigned incode, unsigned runmask, unsigned long long nit, OfxLoopDev *dev,
                  int launch) -> int {
        auto kern = nt_stores ? k_tvl1_iter3<T, true, false> : k_tvl1_iter3<T, false, false>;
        if constexpr (sizeof(T) == sizeof(double)) {
            if (strict) kern = nt_stores ? k_tvl1_iter3<T, true, true> : k_tvl1_iter3<T, false, true>;
        }
        hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, L.Ut(), L.P1t(), L.P2t(), L.A, (const T *) L.R, ctx->d_err, k0, check,
                           slot0, nx, ny, rows, strips_x, strips_pad, l_t, theta, taut, thr, incode, runmask, nit, err_stride, dev, launch,
                           afac1, afac2, S.max_iter);
        OFX_LAUNCH_CHECK(ctx);
        return OFX_OK;
    };
    auto code_of_unit = [&](unsigned unit) -> unsigned {
        unsigned c = 0;
        for (int g = 0; g < G; g++) c |= ((b0[g] + unit) % 3u) << (2 * g);
        return c;
    };
    int unit_of[OFX_MAX_GROUP];
    int inside[OFX_MAX_GROUP];
    if (cursor) {
        // (ctx->d_state is read at launch time: the loop driver may re-allocate the state block before the first launch)
        auto launch = [&](int Lidx, double thr) -> int {
            return go(0, 1, 0, thr, code_of_unit((unsigned) Lidx), all, 0ull, reinterpret_cast<OfxLoopDev *>(ctx->d_state), Lidx);
        };
        auto redo = [&](const int *n_of, const int *k0_of, const int *u_of) -> int {
            unsigned incode = 0, runmask = 0;
            unsigned long long nit = 0;
            for (int g = 0; g < G; g++) {
                if (n_of[g] < 0) continue;
                runmask |= 1u << g;
                incode |= ((b0[g] + (unsigned) u_of[g]) % 3u) << (2 * g);
                nit |= (unsigned long long) (n_of[g] - k0_of[g]) << (4 * g);
            }
            return go(0, 0, S.max_iter, -1.0, incode, runmask, nit, nullptr, 0);    // no stopping test; errors into the scratch slot
        };
        const int upc = (S.chunk + 2) / 3 < 1 ? 1 : (S.chunk + 2) / 3;
        // the cursor loop reserves G * (max_iter + 1) slots itself; make sure the state block it clears is the large one
        OFX_TRY(ofx_run_loop_cursor(ctx, S, G, upc, 3, launch, redo, n_out, err_out, unit_of, inside, ms_out));
    } else {
        auto launch = [&](int k, int cnt, double thr) -> int {
            unsigned long long nit = 0;
            for (int g = 0; g < G; g++) nit |= (unsigned long long) cnt << (4 * g);
            return go(k, 1, k, thr, code_of_unit((unsigned) (k / 3)), all, nit, nullptr, 0);
        };
        auto redo = [&](const int *k_of) -> int {
            unsigned incode = 0, runmask = 0;
            unsigned long long nit = 0;
            for (int g = 0; g < G; g++) {
                if (k_of[g] < 0) continue;
                const int n = k_of[g] + 1, j = (n - 1) / 3;
                runmask |= 1u << g;
                incode |= ((b0[g] + (unsigned) j) % 3u) << (2 * g);
                nit |= (unsigned long long) (n - j * 3) << (4 * g);
            }
            return go(0, 0, S.max_iter, -1.0, incode, runmask, nit, nullptr, 0);
        };
        int took_alt[OFX_MAX_GROUP];
        OFX_TRY(ofx_run_loop_group(ctx, S, G, launch, redo, n_out, err_out, ms_out, 0u, took_alt));
        for (int g = 0; g < G; g++) {
            unit_of[g] = n_out[g] > 0 ? (n_out[g] - 1) / 3 : -1;
            inside[g] = (n_out[g] % 3 != 0 && n_out[g] != S.max_iter) ? 1 : 0;
        }
    }
    unsigned cur = 0;
    for (int g = 0; g < G; g++) {
        const unsigned units = (unsigned) (unit_of[g] + 1);        // the result is the output of the unit that contains iteration n - 1
        cur |= ((b0[g] + units) % 3u) << (2 * g);
        L.last_n[g] = n_out[g];
        if (alt_out) alt_out[g] = inside[g] ? 1 : 0;               // ended inside a unit: its first iterations were re-run
    }
    L.cur = cur;
    return OFX_OK;
}

// src/tvl1flow.cpp:46-212 on device-resident level data; L.U[half of L.cur] holds the incoming flows.
// stats[g] = work record of pair g.
template <typename T>
static int tvl1_single_scale_dev(ofx_ctx *ctx, Tvl1Level<T> &L, const Tvl1Params &P, int scale, ofx_stats *stats)
{
    const int nx = L.nx, ny = L.ny, G = L.G;
    const size_t n = L.n();
    const dim3 g2(ofx_cdiv(nx, OFX_WARP_BX), ofx_cdiv(ny, OFX_WARP_BY), G), b2(OFX_WARP_BX, OFX_WARP_BY);
    if (nx < 2 || ny < 2) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: level %dx%d too small", nx, ny);

    OFX_TRY(op_grad_pack<T>(ctx, L.I1, L.pa, L.pb, nx, ny, G));                                                    // :84
    // p = 0 in the buffer each pair's u lives in (:87-90)
    for (int h = 0; h < 3; h++) {
        bool used = false;
        for (int g = 0; g < G; g++) used = used || (L.curidx(g) == (unsigned) h);
        if (!used) continue;
        OFX_TRY(op_fill2<T>(ctx, L.P1[h], n * G));
        OFX_TRY(op_fill2<T>(ctx, L.P2[h], n * G));
    }
    for (int g = 0; g < G; g++) L.last_n[g] = 0;

    for (int w = 0; w < P.warps; w++) {
        if (ctx->warp_lds)
            hipLaunchKernelGGL(k_tvl1_warp_lds<T>, dim3(ofx_cdiv(nx, 32), ofx_cdiv(ny, WARP_BY), G), dim3(32, WARP_BY), 0, ctx->stream, L.pa,
                               (const T *) L.pb, (const T *) L.I0, L.Ut(), L.A, L.R, nx, ny, L.cur);
        else
            hipLaunchKernelGGL(k_tvl1_warp<T>, g2, b2, 0, ctx->stream, L.pa, (const T *) L.pb, (const T *) L.I0, L.Ut(),
                               L.A, L.R, nx, ny, L.cur);                                                           // :94-109
        OFX_LAUNCH_CHECK(ctx);
        // p lives in the same ping-pong half as u
        int it[OFX_MAX_GROUP];
        double error[OFX_MAX_GROUP];
        float ms = 0.f;
        int odd[OFX_MAX_GROUP], unit = 0;
        OFX_TRY(tvl1_run_iterations<T>(ctx, L, P, it, error, ctx->profile ? &ms : nullptr, odd, &unit));
        if (P.verbose && G == 1) fprintf(stderr, "Warping: %d, Iterations: %d, Error: %f\n", w, it[0], error[0]);   // :184-188
        for (int g = 0; g < G; g++) {
            ofx_stats &S = stats[g];
            S.odd_stops += odd[g] != 0;
            S.odd_stops_stored += odd[g] == 2;
            if (scale < OFX_MAX_SCALES) {
                S.fused[scale] = unit;
                if (w < OFX_MAX_SOLVES) { S.iters[scale][w] = it[g]; S.error[scale][w] = error[g]; }
                S.iter_ms[scale] += ms;
                S.iter_launches[scale] += it[g];
            }
            S.work_pix_iters += (double) it[g] * nx * ny;
        }
    }
    return OFX_OK;
}

static void stats_begin(ofx_stats *st, int nscales, int nsolves)
{
    memset(st, 0, sizeof(*st));
    st->nscales = nscales;
    st->nsolves = nsolves;
}

// src/tvl1flow.cpp:219-328 for G pairs in lockstep.  dI0[g] / dI1[g]: device images of storage type T.  On
// success lv[0].U[half of lv[0].cur] holds the flows.
template <typename T>
static int tvl1_multiscale_dev(ofx_ctx *ctx, int G, const T *const *dI0, const T *const *dI1, int nxx, int nyy,
                               const Tvl1Params &P, int nscales, double zfactor, std::vector<Tvl1Level<T>> &lv,
                               ofx_stats *stats)
{
    if (P.warps < 1) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: warps=%d", P.warps);
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: group of %d pairs", G);

    std::vector<int> nxs, nys;
    OFX_TRY(op_pyramid_sizes(ctx, nxx, nyy, nscales, zfactor, nxs, nys));
    for (int g = 0; g < G; g++) {
        stats_begin(&stats[g], nscales, P.warps);
        for (int s = 0; s < nscales && s < OFX_MAX_SCALES; s++) { stats[g].nx[s] = nxs[s]; stats[g].ny[s] = nys[s]; }
    }
    lv.resize(nscales);
    for (int s = 0; s < nscales; s++) OFX_TRY(tvl1_level_alloc<T>(ctx, lv[s], nxs[s], nys[s], G, true));
    {                                                                                              // :255-275
        T *tmpA, *tmpB;
        double *scr;
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nxx * nyy, &tmpA));
        OFX_TRY(ofx_alloc(ctx, (size_t) 2 * G * nxx * nyy, &tmpB));
        OFX_TRY(ofx_alloc(ctx, (size_t) G * op_pyramid_scratch_doubles(), &scr));
        std::vector<T *> lA(nscales), lB(nscales);
        for (int s = 0; s < nscales; s++) { lA[s] = lv[s].I0; lB[s] = lv[s].I1; }
        OFX_TRY(op_build_pyramid_group<T>(ctx, G, (const void *const *) dI0, (const void *const *) dI1, nscales, zfactor,
                                          TVL1_PRESMOOTHING_SIGMA, nxs.data(), nys.data(), lA.data(), lB.data(), tmpA, tmpB, scr));
    }
    Tvl1Level<T> &C = lv[nscales - 1];
    OFX_TRY(op_fill2<T>(ctx, C.U[0], C.n() * G));                                                // :278-280

    for (int s = nscales - 1; s >= 0; s--) {                                                     // :283
        if (P.verbose && G == 1) fprintf(stderr, "Scale %d: %dx%d\n", s, lv[s].nx, lv[s].ny);
        OFX_TRY(tvl1_single_scale_dev<T>(ctx, lv[s], P, s, stats));
        if (!s) break;
        lv[s - 1].cur = 0;
        {                                                                                        // :302-309, all pairs in one launch
            const int nxf = lv[s - 1].nx, nyf = lv[s - 1].ny;
            hipLaunchKernelGGL(k_tvl1_zoom_in_g<T>, dim3(ofx_cdiv(nxf, 64), ofx_cdiv(nyf, 4), G), dim3(64, 4), 0, ctx->stream,
                               lv[s].Ut(), lv[s].cur, lv[s - 1].U[0], lv[s].nx, lv[s].ny, nxf, nyf, (double) nxf / lv[s].nx,
                               (double) nyf / lv[s].ny, 1.0 / zfactor);
            OFX_LAUNCH_CHECK(ctx);
        }
    }
    return OFX_OK;
}

static Tvl1Params make_params(const ofx_ctx *ctx, double tau, double lambda, double theta, int warps,
                               double epsilon, int verbose)
{
    Tvl1Params P;
    P.tau = tau; P.lambda = lambda; P.theta = theta; P.epsilon = epsilon;
    P.warps = warps; P.verbose = verbose;
    P.max_iter = OFX_TVL1_MAX_ITERATIONS;
    P.fixed = ctx->fixed_work != 0;     // option "fixed_work": every warp runs exactly MAX_ITERATIONS
    return P;
}

template <typename T>
static int upload_image(ofx_ctx *ctx, const double *h, size_t n, T **out)
{
    double *stage;
    OFX_TRY(ofx_alloc(ctx, n, &stage));
    OFX_TRY(ofx_alloc(ctx, n, out));
    OFX_HIP(ctx, hipMemcpyAsync(stage, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return op_convert_in<T>(ctx, stage, *out, n);
}

template <typename T>
static int tvl1_multiscale_host(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2, int nx,
                                int ny, const Tvl1Params &P, int nscales, double zfactor)
{
    const size_t n = (size_t) nx * ny;
    T *dI0, *dI1;
    OFX_TRY(upload_image<T>(ctx, I0, n, &dI0));
    OFX_TRY(upload_image<T>(ctx, I1, n, &dI1));
    std::vector<Tvl1Level<T>> lv;
    const T *a = dI0, *b = dI1;
    OFX_TRY(tvl1_multiscale_dev<T>(ctx, 1, &a, &b, nx, ny, P, nscales, zfactor, lv, &ctx->stats));
    double *d1, *d2;
    OFX_TRY(ofx_alloc(ctx, n, &d1));
    OFX_TRY(ofx_alloc(ctx, n, &d2));
    OFX_TRY(op_deinterleave2<T>(ctx, lv[0].Ucur(0), d1, d2, n));
    OFX_HIP(ctx, hipMemcpyAsync(u1, d1, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(u2, d2, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

extern "C" int ofx_tvl1_multiscale(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2, int nx,
                                   int ny, double tau, double lambda, double theta, int nscales, double zfactor,
                                   int warps, double epsilon, int verbose)
{
    OFX_ENTER(ctx);
    if (!I0 || !I1 || !u1 || !u2) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: NULL pointer");
    const double t0 = ofx_now_ms();
    const Tvl1Params P = make_params(ctx, tau, lambda, theta, warps, epsilon, verbose);
    int s = ctx->precision == OFX_F64
                ? tvl1_multiscale_host<double>(ctx, I0, I1, u1, u2, nx, ny, P, nscales, zfactor)
                : tvl1_multiscale_host<float>(ctx, I0, I1, u1, u2, nx, ny, P, nscales, zfactor);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

// G device-resident pairs in lockstep; stats = G records
// Is p device memory of this context's GPU?  Everything else -- another GPU's memory, pinned or pageable host memory -- is
// reached through a staging copy (hipMemcpyDefault resolves the direction; between GPUs it is a peer copy over xGMI).
static bool ofx_ptr_is_local(const ofx_ctx *ctx, const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void) hipGetLastError();                              // pageable host memory is unknown to the runtime
        return false;
    }
    return a.type == hipMemoryTypeDevice && a.device == ctx->device;
}
// How a buffer that lives on `mem_device` (-1: host memory) reaches a context on `ctx_device`: 0 = in place, 1 = one
// hipMemcpyDefault (host <-> device, or a peer copy over xGMI), 2 = two copies through a host bounce buffer because the two GPUs
// cannot address each other (hipDeviceCanAccessPeer says no: another IOMMU group, peer access disabled, ...).  Pure decision,
// exported so that the host-logic tests cover it without a GPU.
extern "C" int ofx_staging_route(int ctx_device, int mem_device, int can_access_peer)
{
    if (mem_device == ctx_device) return 0;
    if (mem_device < 0) return 1;
    return can_access_peer ? 1 : 2;
}
// dst <- src (one of them on this context's GPU, the other anywhere), on the context's stream.  A route-2 copy is synchronous
// and leaves a note in ofx_last_error (the call still succeeds).  UNEXECUTED on more than one GPU (no multi-GPU box in any round).
static int ofx_copy_staged(ofx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    const void *other = ofx_ptr_is_local(ctx, dst) ? src : dst;
    int mem_device = -1, can = 1;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, other) == hipSuccess) {
        if (a.type == hipMemoryTypeDevice) mem_device = a.device;
    } else {
        (void) hipGetLastError();
    }
    if (mem_device >= 0 && mem_device != ctx->device && hipDeviceCanAccessPeer(&can, ctx->device, mem_device) != hipSuccess) {
        (void) hipGetLastError();
        can = 0;
    }
    if (ofx_staging_route(ctx->device, mem_device, can) < 2) {
        OFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
        return OFX_OK;
    }
    void *bounce = nullptr;
    OFX_HIP(ctx, hipHostMalloc(&bounce, bytes, hipHostMallocDefault));
    hipError_t e = hipStreamSynchronize(ctx->stream);                                       // src may be this stream's product
    if (e == hipSuccess) e = hipMemcpy(bounce, src, bytes, hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(dst, bounce, bytes, hipMemcpyDefault);
    (void) hipHostFree(bounce);
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "staging through the host failed: %s", hipGetErrorString(e));
    snprintf(ctx->errmsg, sizeof(ctx->errmsg), "note: GPU %d cannot address GPU %d, buffers were staged through the host", ctx->device,
             mem_device);
    return OFX_OK;
}

template <typename T>
static int tvl1_group_devapi(ofx_ctx *ctx, int G, const void *const *dI0_in, const void *const *dI1_in, void *const *d_flo_in,
                             int nx, int ny, const Tvl1Params &P, int nscales, double zfactor, ofx_stats *stats)
{
    const size_t n = (size_t) nx * ny;
    // Images and payloads that do not live on this context's GPU (a batch spread over several GPUs by ofx_tvl1_batch_dev, or
    // host buffers) are staged: in through arena buffers before the solve, out after it, all on the context's stream.
    const void *dI0[OFX_MAX_GROUP], *dI1[OFX_MAX_GROUP];
    void *d_flo[OFX_MAX_GROUP];
    for (int g = 0; g < G; g++) {
        const void *src[2] = {dI0_in[g], dI1_in[g]};
        const void **dst[2] = {&dI0[g], &dI1[g]};
        for (int k = 0; k < 2; k++) {
            if (ofx_ptr_is_local(ctx, src[k])) { *dst[k] = src[k]; continue; }
            T *stage;
            OFX_TRY(ofx_alloc(ctx, n, &stage));
            OFX_TRY(ofx_copy_staged(ctx, stage, src[k], n * sizeof(T)));
            *dst[k] = stage;
        }
        d_flo[g] = d_flo_in[g];
        if (!ofx_ptr_is_local(ctx, d_flo_in[g])) {
            float2 *stage;
            OFX_TRY(ofx_alloc(ctx, n, &stage));
            d_flo[g] = stage;
        }
    }
    std::vector<Tvl1Level<T>> lv;
    OFX_TRY(tvl1_multiscale_dev<T>(ctx, G, (const T *const *) dI0, (const T *const *) dI1, nx, ny, P, nscales, zfactor, lv,
                                   stats));
    OfxGroupPtrs out;
    for (int g = 0; g < OFX_MAX_GROUP; g++) { out.a[g] = g < G ? d_flo[g] : nullptr; out.b[g] = nullptr; }
    hipLaunchKernelGGL(k_tvl1_to_flo_g<T>, dim3((unsigned) ((n + 255) / 256), G), dim3(256), 0, ctx->stream, lv[0].Ut(), lv[0].cur,
                       out, n);
    OFX_LAUNCH_CHECK(ctx);
    for (int g = 0; g < G; g++)
        if (d_flo[g] != d_flo_in[g]) OFX_TRY(ofx_copy_staged(ctx, d_flo_in[g], d_flo[g], n * sizeof(float2)));
    return OFX_OK;
}

extern "C" int ofx_tvl1_multiscale_dev(ofx_ctx *ctx, const void *dI0, const void *dI1, void *d_flo, int nx, int ny,
                                       double tau, double lambda, double theta, int nscales, double zfactor,
                                       int warps, double epsilon, int verbose)
{
    OFX_ENTER(ctx);
    if (!dI0 || !dI1 || !d_flo) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: NULL pointer");
    const double t0 = ofx_now_ms();
    const Tvl1Params P = make_params(ctx, tau, lambda, theta, warps, epsilon, verbose);
    int s = ctx->precision == OFX_F64
                ? tvl1_group_devapi<double>(ctx, 1, &dI0, &dI1, &d_flo, nx, ny, P, nscales, zfactor, &ctx->stats)
                : tvl1_group_devapi<float>(ctx, 1, &dI0, &dI1, &d_flo, nx, ny, P, nscales, zfactor, &ctx->stats);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

extern "C" int ofx_tvl1_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI0, const void *const *dI1,
                                  void *const *d_flo, int nx, int ny, double tau, double lambda, double theta,
                                  int nscales, double zfactor, int warps, double epsilon, ofx_stats *stats_out)
{
    OFX_ENTER(ctx);
    if (!dI0 || !dI1 || !d_flo) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: NULL pointer");
    if (n_pairs < 1 || n_pairs > OFX_MAX_GROUP)
        return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: a lockstep group holds 1..%d pairs (got %d)", OFX_MAX_GROUP, n_pairs);
    for (int g = 0; g < n_pairs; g++)
        if (!dI0[g] || !dI1[g] || !d_flo[g]) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: NULL pointer (pair %d)", g);
    const double t0 = ofx_now_ms();
    const Tvl1Params P = make_params(ctx, tau, lambda, theta, warps, epsilon, 0);
    std::vector<ofx_stats> local(stats_out ? 0 : n_pairs);      // ~25 KB per record: not on the stack
    ofx_stats *st = stats_out ? stats_out : local.data();
    int s = ctx->precision == OFX_F64
                ? tvl1_group_devapi<double>(ctx, n_pairs, dI0, dI1, d_flo, nx, ny, P, nscales, zfactor, st)
                : tvl1_group_devapi<float>(ctx, n_pairs, dI0, dI1, d_flo, nx, ny, P, nscales, zfactor, st);
    const double ms = ofx_now_ms() - t0;
    for (int g = 0; g < n_pairs; g++) st[g].total_ms = ms;
    ctx->stats = st[0];
    return s;
}

template <typename T>
static int tvl1_single_scale_host(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2, int nx,
                                  int ny, const Tvl1Params &P)
{
    const size_t n = (size_t) nx * ny;
    stats_begin(&ctx->stats, 1, P.warps);
    ctx->stats.nx[0] = nx;
    ctx->stats.ny[0] = ny;
    Tvl1Level<T> L;
    OFX_TRY(tvl1_level_alloc<T>(ctx, L, nx, ny, 1, false));
    OFX_TRY(upload_image<T>(ctx, I0, n, &L.I0));
    OFX_TRY(upload_image<T>(ctx, I1, n, &L.I1));
    double *d1, *d2;
    OFX_TRY(ofx_alloc(ctx, n, &d1));
    OFX_TRY(ofx_alloc(ctx, n, &d2));
    OFX_HIP(ctx, hipMemcpyAsync(d1, u1, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(d2, u2, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_TRY(op_interleave2<T>(ctx, d1, d2, L.U[0], n));
    OFX_TRY(tvl1_single_scale_dev<T>(ctx, L, P, 0, &ctx->stats));
    OFX_TRY(op_deinterleave2<T>(ctx, L.Ucur(0), d1, d2, n));
    OFX_HIP(ctx, hipMemcpyAsync(u1, d1, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(u2, d2, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

extern "C" int ofx_tvl1_single_scale(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2,
                                     int nx, int ny, double tau, double lambda, double theta, int warps,
                                     double epsilon, int verbose)
{
    OFX_ENTER(ctx);
    if (!I0 || !I1 || !u1 || !u2) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: NULL pointer");
    if (nx < 2 || ny < 2 || warps < 1) return ofx_fail(ctx, OFX_ERR_ARG, "tvl1: bad size / warps");
    const double t0 = ofx_now_ms();
    const Tvl1Params P = make_params(ctx, tau, lambda, theta, warps, epsilon, verbose);
    int s = ctx->precision == OFX_F64 ? tvl1_single_scale_host<double>(ctx, I0, I1, u1, u2, nx, ny, P)
                                      : tvl1_single_scale_host<float>(ctx, I0, I1, u1, u2, nx, ny, P);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

template <typename T>
static int tvl1_iterations_host(ofx_ctx *ctx, double *u1, double *u2, double *p11, double *p12, double *p21,
                                double *p22, const double *I1wx, const double *I1wy, const double *rho_c, int nx,
                                int ny, const Tvl1Params &P, double *error)
{
    const size_t n = (size_t) nx * ny;
    stats_begin(&ctx->stats, 1, 1);
    ctx->stats.nx[0] = nx;
    ctx->stats.ny[0] = ny;
    Tvl1Level<T> L;
    OFX_TRY(tvl1_level_alloc<T>(ctx, L, nx, ny, 1, false));
    double *d[2];
    OFX_TRY(ofx_alloc(ctx, n, &d[0]));
    OFX_TRY(ofx_alloc(ctx, n, &d[1]));
    struct { const double *a, *b; typename Pix<T>::v2 *dst; } up[4] = {
        {u1, u2, L.U[0]}, {p11, p12, L.P1[0]}, {p21, p22, L.P2[0]}, {I1wx, I1wy, L.A}};
    for (auto &e : up) {
        OFX_HIP(ctx, hipMemcpyAsync(d[0], e.a, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        OFX_HIP(ctx, hipMemcpyAsync(d[1], e.b, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        OFX_TRY(op_interleave2<T>(ctx, d[0], d[1], e.dst, n));
    }
    OFX_HIP(ctx, hipMemcpyAsync(d[0], rho_c, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_TRY(op_convert_in<T>(ctx, d[0], L.R, n));

    int it = 0;
    double err = 0.0;
    float ms = 0.f;
    OFX_TRY(tvl1_run_iterations<T>(ctx, L, P, &it, &err, ctx->profile ? &ms : nullptr));
    ctx->stats.iters[0][0] = it;
    ctx->stats.error[0][0] = err;
    ctx->stats.iter_ms[0] = ms;
    ctx->stats.iter_launches[0] = it;
    ctx->stats.work_pix_iters = (double) it * nx * ny;
    if (error) *error = err;

    struct { double *a, *b; typename Pix<T>::v2 *src; } down[3] = {
        {u1, u2, L.Ucur(0)}, {p11, p12, L.P1cur(0)}, {p21, p22, L.P2cur(0)}};
    for (auto &e : down) {
        OFX_TRY(op_deinterleave2<T>(ctx, e.src, d[0], d[1], n));
        OFX_HIP(ctx, hipMemcpyAsync(e.a, d[0], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        OFX_HIP(ctx, hipMemcpyAsync(e.b, d[1], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

extern "C" int ofx_tvl1_iterations(ofx_ctx *ctx, double *u1, double *u2, double *p11, double *p12, double *p21,
                                   double *p22, const double *I1wx, const double *I1wy, const double *rho_c, int nx,
                                   int ny, double tau, double lambda, double theta, int n_iter, double *error)
{
    OFX_ENTER(ctx);
    if (!u1 || !u2 || !p11 || !p12 || !p21 || !p22 || !I1wx || !I1wy || !rho_c)
        return ofx_fail(ctx, OFX_ERR_ARG, "tvl1_iterations: NULL pointer");
    if (nx < 2 || ny < 2 || n_iter < 1 || n_iter > OFX_TVL1_MAX_ITERATIONS)
        return ofx_fail(ctx, OFX_ERR_ARG, "tvl1_iterations: bad size / n_iter");
    const double t0 = ofx_now_ms();
    Tvl1Params P = make_params(ctx, tau, lambda, theta, 1, 0.0, 0);
    P.max_iter = n_iter;
    P.fixed = true;
    int s = ctx->precision == OFX_F64
                ? tvl1_iterations_host<double>(ctx, u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, nx, ny, P, error)
                : tvl1_iterations_host<float>(ctx, u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, nx, ny, P, error);
    ctx->stats.total_ms = ofx_now_ms() - t0;
    return s;
}

// Pairs per lockstep group that ofx_tvl1_batch_dev uses for this batch.  The "lockstep" option of ctxs[0] if
// set; otherwise the batch is spread over as few rounds as the largest allowed group permits (every context
// gets one group per round) and the groups are evened out: 20 pairs on 4 contexts = 4 groups of 5, not
// 5 groups of 4 with one context working alone at the end.  Measured on MI355X (1080p f64, 4 contexts):
// bigger groups keep winning up to the cap (32 pairs: groups of 4 / 8 = 48.9k / 50.6k Mpix*it/s; 64 pairs:
// 4 / 8 / 16 = 49.5k / 50.8k / 51.2k).  The cap is OFX_MAX_GROUP, lowered so that all contexts' level arrays
// together stay within half of the device memory that is free now (or within option "mem_budget" bytes).
extern "C" int ofx_tvl1_batch_group_size(ofx_ctx *const *ctxs, int n_ctx, int n_pairs, int nx, int ny, int nscales,
                                         double zfactor)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0] || n_pairs < 0 || nx < 1 || ny < 1) return -OFX_ERR_ARG;
    int G = ctxs[0]->lockstep;
    if (G > 0) return G > OFX_MAX_GROUP ? OFX_MAX_GROUP : G;
    if (n_pairs <= 1) return 1;
    std::vector<int> nxs, nys;
    const int st = op_pyramid_sizes(ctxs[0], nx, ny, nscales, zfactor, nxs, nys);
    if (st != OFX_OK) return -st;
    double px = 0;
    for (int s = 0; s < nscales; s++) px += (double) nxs[s] * nys[s];
    const double elem = ctxs[0]->precision == OFX_F32 ? 4.0 : 8.0;
    // tvl1_level_alloc: I0, I1, pb, R (1 element per pixel each) + pa, U[3], P1[3], P2[3], A (2 each) = 26 per level
    // + the pyramid temporaries: two scratch arrays of 2 full-size images per pair (op_build_pyramid_group)
    const double per_pair = (26.0 * px + 4.0 * (double) nx * ny) * elem;
    const double per_ctx = 64e6;                                    // error slots, arena slack
    // the budget is per DEVICE: the contexts of a batch may sit on several GPUs (ofx_tvl1_batch_dev), each of which holds the level
    // arrays of its own contexts only -- free memory of every distinct device, divided by the contexts on it; the tightest fit wins
    int dev_now = 0;
    (void) hipGetDevice(&dev_now);
    int cap = OFX_MAX_GROUP;
    for (int w = 0; w < n_ctx; w++) {
        if (!ctxs[w]) return -OFX_ERR_ARG;
        int on_dev = 0;
        bool seen = false;
        for (int v = 0; v < n_ctx; v++) {
            if (!ctxs[v] || ctxs[v]->device != ctxs[w]->device) continue;
            on_dev++;
            seen = seen || v < w;
        }
        if (seen) continue;                                         // this device was handled with its first context
        double budget = ctxs[w]->mem_budget > 0 ? ctxs[w]->mem_budget * on_dev / n_ctx : -1.0;    // "mem_budget": all contexts of the batch
        if (budget < 0) {
            size_t mfree = 0, mtotal = 0;
            (void) hipSetDevice(ctxs[w]->device);
            if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) budget = 0.5 * (double) mfree;
            else (void) hipGetLastError();
        }
        if (budget >= 0) {
            const double fit = (budget / on_dev - per_ctx) / per_pair;
            if (fit < cap) cap = fit < 1.0 ? 1 : (int) fit;
        }
    }
    (void) hipSetDevice(dev_now);
    const int rounds = (n_pairs + n_ctx * cap - 1) / (n_ctx * cap);
    G = (n_pairs + n_ctx * rounds - 1) / (n_ctx * rounds);
    return G < 1 ? 1 : G;
}

// libm's hypot as the dual update uses it (src/tvl1flow.cpp:172-173), n values: the device restatement of glibc's algorithm
// exposed for direct testing over the whole double range
__global__ void k_hypot(const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hypot_ref(x[i], y[i]);
}
extern "C" int ofx_hypot(ofx_ctx *ctx, const double *x, const double *y, double *out, int n)
{
    OFX_ENTER(ctx);
    if (!x || !y || !out || n < 1) return ofx_fail(ctx, OFX_ERR_ARG, "hypot: NULL pointer / n < 1");
    double *dx, *dy, *dz;
    OFX_TRY(ofx_alloc(ctx, (size_t) n, &dx));
    OFX_TRY(ofx_alloc(ctx, (size_t) n, &dy));
    OFX_TRY(ofx_alloc(ctx, (size_t) n, &dz));
    OFX_HIP(ctx, hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_hypot, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const double *) dx, (const double *) dy, dz, n);
    OFX_LAUNCH_CHECK(ctx);
    OFX_HIP(ctx, hipMemcpyAsync(out, dz, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

// ---- batch of pairs: lockstep groups, one worker thread per context --------------------------------------
// The pairs are cut into groups of ofx_tvl1_batch_group_size() consecutive pairs; group q is
// solved by context q % n_ctx with ofx_tvl1_group_dev, i.e. its pairs share every launch.  On the small
// pyramid levels a launch is a latency chain that leaves most of the GPU idle, so G pairs per launch cost
// the time of one; and the convergence polls (one host round trip per warp) are paid once per group.
extern "C" int ofx_tvl1_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI0, const void *const *dI1,
                                  void *const *d_flo, int n_pairs, int nx, int ny, double tau, double lambda,
                                  double theta, int nscales, double zfactor, int warps, double epsilon,
                                  double *work_pix_iters)
{
    if (!ctxs || n_ctx < 1 || n_pairs < 0 || !dI0 || !dI1 || !d_flo) return OFX_ERR_ARG;
    // The contexts may live on DIFFERENT GPUs of the node (one process, one host thread per context): a group whose images
    // are not on its context's GPU is staged there and its payloads are copied back into the caller's arrays (tvl1_group_devapi)
    // -- the C caller's multi-GPU path, no launcher and no collective library needed (pairs are independent).
    for (int w = 0; w < n_ctx; w++)
        if (!ctxs[w] || ctxs[w]->precision != ctxs[0]->precision) return OFX_ERR_ARG;
    const int G = ofx_tvl1_batch_group_size(ctxs, n_ctx, n_pairs, nx, ny, nscales, zfactor);
    if (G < 1) return G < 0 ? -G : OFX_ERR_ARG;                 // negative = -status of the failing check
    const int n_groups = (n_pairs + G - 1) / G;
    std::atomic<int> status(OFX_OK);
    auto worker = [&](int w) {
        std::vector<ofx_stats> st(G);
        for (int q = w; q < n_groups; q += n_ctx) {
            if (status.load() != OFX_OK) return;
            const int first = q * G, cnt = (n_pairs - first < G) ? n_pairs - first : G;
            const int s = ofx_tvl1_group_dev(ctxs[w], cnt, dI0 + first, dI1 + first, d_flo + first, nx, ny, tau, lambda, theta,
                                             nscales, zfactor, warps, epsilon, st.data());
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
    return status.load();
}
