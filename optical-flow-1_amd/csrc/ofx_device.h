// ofx_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Arithmetic contract: every value is widened to double on load, all arithmetic is IEEE double in
// the reference's association order (the library is built with -ffp-contract=off: no FMA
// contraction), and results are rounded once on store to the storage type T (double: no-op).
#pragma once

#include <hip/hip_runtime.h>

#define OFX_DEV __device__ __forceinline__

// ---- widening loads / narrowing stores --------------------------------------------------------
OFX_DEV double  ldw(const double *p) { return *p; }
OFX_DEV double  ldw(const float *p)  { return (double) *p; }
OFX_DEV double2 ldw2(const double2 *p) { return *p; }
OFX_DEV double2 ldw2(const float2 *p)  { float2 v = *p; return make_double2((double) v.x, (double) v.y); }
OFX_DEV double4 ldw4(const double4 *p) { return *p; }
OFX_DEV double4 ldw4(const float4 *p)
{
    float4 v = *p;
    return make_double4((double) v.x, (double) v.y, (double) v.z, (double) v.w);
}
OFX_DEV void stn(double *p, double v) { *p = v; }
OFX_DEV void stn(float *p, double v)  { *p = (float) v; }
OFX_DEV void stn2(double2 *p, double2 v) { *p = v; }
OFX_DEV void stn2(float2 *p, double2 v)  { *p = make_float2((float) v.x, (float) v.y); }
// Non-temporal stores for results that no kernel will find in cache again (working set of the level
// larger than the 256 MiB Infinity Cache): measured +3.7 % on the 4K iteration kernel, -3 % at 1080p
// where the ping-pong output of one launch is still resident when the next launch reads it.
typedef double ofx_d2v __attribute__((ext_vector_type(2)));
typedef float ofx_f2v __attribute__((ext_vector_type(2)));
OFX_DEV void stn2_nt(double2 *p, double2 v)
{
    ofx_d2v w;
    w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<ofx_d2v *>(p));
}
OFX_DEV void stn2_nt(float2 *p, double2 v)
{
    ofx_f2v w;
    w.x = (float) v.x; w.y = (float) v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<ofx_f2v *>(p));
}
template <bool NT, typename V> OFX_DEV void stn2_sel(V *p, double2 v)
{
    if (NT) stn2_nt(p, v);
    else stn2(p, v);
}
// Predicated stores without control flow.  A store inside `if (owner) ...` sits in its own basic block, and
// the compiler's s_waitcnt placement then has to assume at the top of a marching loop that NO store was
// issued after the prefetch loads -- it waits with vmcnt(0), i.e. for the stores of the previous step as
// well, and the whole write latency (~1.4 us per marching step) lands on the critical path of a wave.
// Buffer stores drop lanes whose offset is outside the descriptor's range, so the predicate becomes part
// of the address (OFX_OOB for lanes that must not write), the instruction is issued unconditionally and
// the loads are awaited with an exact vmcnt(#stores).
#define OFX_OOB 0xFFFFFFF0u
typedef unsigned ofx_u4v __attribute__((ext_vector_type(4)));
typedef unsigned ofx_u2v __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t ofx_rsrc;
// raw buffer over [p, p + bytes): stride 0, 32-bit data format (descriptor word 3 of the gfx9 family)
OFX_DEV ofx_rsrc make_rsrc(void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000); }
template <bool NT> OFX_DEV void bst2(ofx_rsrc r, unsigned byte_off, double2 v, const double2 *)
{
    ofx_u4v w;
    __builtin_memcpy(&w, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, byte_off, 0, NT ? 2 : 0);
}
template <bool NT> OFX_DEV void bst2(ofx_rsrc r, unsigned byte_off, double2 v, const float2 *)
{
    const float2 f = make_float2((float) v.x, (float) v.y);
    ofx_u2v w;
    __builtin_memcpy(&w, &f, 8);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, byte_off, 0, NT ? 2 : 0);
}
OFX_DEV void stn4(double4 *p, double4 v) { *p = v; }
OFX_DEV void stn4(float4 *p, double4 v)
{
    *p = make_float4((float) v.x, (float) v.y, (float) v.z, (float) v.w);
}

// Loads of unknowns that other waves of the SAME workgroup rewrite inside one launch (windowed SOR sweeps):
// COH = true bypasses the vector L1 (agent-scope relaxed atomic loads = global_load ... sc1, served by L2), so a
// line cached before a neighbour's store can never be returned stale.  8-byte pieces: a pixel's pair is written
// by one thread in one store, and the protocol never reads a pixel in the step it is written.
template <bool COH> OFX_DEV double2 ldu2(const double2 *p)
{
    if (!COH) return *p;
    const double *q = reinterpret_cast<const double *>(p);
    double2 r;
    r.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}
template <bool COH> OFX_DEV double2 ldu2(const float2 *p)
{
    if (!COH) { const float2 v = *p; return make_double2((double) v.x, (double) v.y); }
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    float2 v;
    __builtin_memcpy(&v, &w, 8);
    return make_double2((double) v.x, (double) v.y);
}

// ---- wave64 primitives --------------------------------------------------------------------------
// Neighbour-lane moves.  gfx950 (GFX9 family) has whole-wavefront DPP shifts: one v_mov_b32_dpp per
// dword, no LDS crossbar round trip and no address arithmetic (a __shfl compiles to ds_bpermute_b32 +
// lane-index math).  wave_shr:1 moves data towards higher lanes (lane i reads lane i-1), wave_shl:1
// the other way; the lane with no source keeps its own value (bound_ctrl off, old = src).
#ifndef OFX_NO_DPP
OFX_DEV double wave_shift_up(double v)      // value of lane-1 (lane 0 keeps its own)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
OFX_DEV double wave_shift_down(double v)    // value of lane+1 (lane 63 keeps its own)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#else
OFX_DEV double wave_shift_up(double v) { return __shfl_up(v, 1, 64); }
OFX_DEV double wave_shift_down(double v) { return __shfl_down(v, 1, 64); }
#endif
// Butterfly all-reduce: every lane ends with the same bits (each level adds the same two partial
// sums in every lane, only the operand order differs and + is commutative).
OFX_DEV double wave_allreduce_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
OFX_DEV double wave_allreduce_min(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { double o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
    return v;
}
OFX_DEV double wave_allreduce_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { double o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}

// ---- hypot --------------------------------------------------------------------------------------
// The reference calls libm hypot() (src/tvl1flow.cpp:172-173).  glibc >= 2.35 (the image's libm)
// computes it as a correctly-rounded sqrt(ax^2+ay^2) followed by one Newton-style correction
// (Borges' algorithm, non-FMA branch); a bare sqrt(x*x+y*y) differs from it in the last bit for
// ~12 % of inputs.  This is the same sequence of IEEE operations (checked against libm on 2e7
// inputs, tests/test_host_logic.py), so the dual update stays bit-compatible.
OFX_DEV double hypot_kernel(double ax, double ay)
{
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

// The compiler's own expansions of the f64 square root and quotient WITHOUT their range scaling (v_ldexp of tiny arguments
// around the rsq iteration; v_div_scale / v_div_fmas / v_div_fixup around the rcp iteration): the same instruction
// sequence, hence the same bits, wherever that scaling is the identity -- x in [2^-767, 2^1023]; numerator 0 or of
// magnitude >= 2^-968 with a normal quotient, denominator in [2^-1000, 2^1000], and the sign of a zero quotient not needed.
OFX_DEV double sqrt_unscaled(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
OFX_DEV double rcp_newton(double d)              // 1 / d to <= 1 ulp for normal d well inside the exponent range
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
OFX_DEV double div_unscaled(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}
// hypot_kernel on the common domain ax <= 2^511, ay >= 2^-383 (checked by the caller): ax^2 + ay^2 lies in [2^-766, 2^1023],
// h in [2^-383, 2^512]; the correction's numerator t1 + t2 is an exact-product residue, 0 or >= 2^-105 h^2 >= 2^-871 in
// magnitude, its quotient by 2 h is normal, and h - (+-0) = h either way -- so both unscaled expansions apply.
OFX_DEV double hypot_kernel_common(double ax, double ay)
{
    double h = sqrt_unscaled(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= div_unscaled(t1 + t2, 2.0 * h);
    return h;
}

OFX_DEV double hypot_ref(double x, double y)
{
    x = fabs(x);
    y = fabs(y);
    const double ax = x < y ? y : x;
    const double ay = x < y ? x : y;
    const double SCALE = 0x1p-600, LARGE = 0x1p+511, TINY = 0x1p-459, EPS = 0x1p-54;
#ifndef OFX_HYPOT_GENERAL_ONLY
    // The common domain first (flow gradients are never 1e-115 or 1e153): there the general code below takes neither
    // scaling branch and reduces to exactly these two lines.  Everything else -- zeros included -- goes the general way.
    if (ax <= LARGE && ay >= 0x1p-383) {
        if (ax >= ay / EPS) return ax + ay;
        return hypot_kernel_common(ax, ay);
    }
#endif
    if (ax > LARGE) {
        if (ay <= ax * EPS) return ax + ay;
        return hypot_kernel(ax * SCALE, ay * SCALE) / SCALE;
    }
    if (ay < TINY) {
        if (ax >= ay / EPS) return ax + ay;
        return hypot_kernel(ax / SCALE, ay / SCALE) * SCALE;
    }
    if (ax >= ay / EPS) return ax + ay;
    return hypot_kernel(ax, ay);
}

// ---- bicubic sampling (src/bicubic_interpolation.cpp:24-39,108-245, Neumann boundary) -----------
struct BicubicTaps {
    int    col[4];   // mx, x, dx, ddx   (clamped)
    int    row[4];   // my, y, dy, ddy   (clamped)
    double fx, fy;   // offsets against the CLAMPED base index (:243)
    bool   out;      // some tap was clamped (:24-39)
};

OFX_DEV int bc_clamp(int x, int n, bool &out)
{
    if (x < 0)  { out = true; return 0; }
    if (x >= n) { out = true; return n - 1; }
    return x;
}

// Quirks kept on purpose: (int) truncation toward zero (:170); tap direction follows the sign of the
// coordinate (:162-163); `my` is offset by sx, not sy (:173).
OFX_DEV BicubicTaps bicubic_taps(double uu, double vv, int nx, int ny)
{
    BicubicTaps t;
    const int sx = (uu < 0) ? -1 : 1;
    const int sy = (vv < 0) ? -1 : 1;
    const int iu = (int) uu, iv = (int) vv;
    bool out = false;
    t.col[1] = bc_clamp(iu, nx, out);
    t.row[1] = bc_clamp(iv, ny, out);
    t.col[0] = bc_clamp(iu - sx, nx, out);
    t.row[0] = bc_clamp(iv - sx, ny, out);
    t.col[2] = bc_clamp(iu + sx, nx, out);
    t.row[2] = bc_clamp(iv + sy, ny, out);
    t.col[3] = bc_clamp(iu + 2 * sx, nx, out);
    t.row[3] = bc_clamp(iv + 2 * sy, ny, out);
    t.out = out;
    t.fx = uu - t.col[1];
    t.fy = vv - t.row[1];
    return t;
}

// Keys cubic (a = -1/2) in the reference's Horner form (:108-123)
OFX_DEV double cubic_cell(double v0, double v1, double v2, double v3, double x)
{
    return v1 + 0.5 * x * (v2 - v0 + x * (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3 + x * (3.0 * (v1 - v2) + v3 - v0)));
}

// single-channel sample from a planar image: 4 cubics along y (one per tap column), then one in x
template <typename T>
OFX_DEV double bicubic_sample(const T *in, const BicubicTaps &t, int nx)
{
    double c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double v0 = ldw(in + (size_t) t.row[0] * nx + t.col[k]);
        const double v1 = ldw(in + (size_t) t.row[1] * nx + t.col[k]);
        const double v2 = ldw(in + (size_t) t.row[2] * nx + t.col[k]);
        const double v3 = ldw(in + (size_t) t.row[3] * nx + t.col[k]);
        c[k] = cubic_cell(v0, v1, v2, v3, t.fy);
    }
    return cubic_cell(c[0], c[1], c[2], c[3], t.fx);
}

// Three-channel sample (f, fx, fy) for the warps: the channels live in a pair plane pa = (f, fx) and a scalar
// plane pb = fy instead of one padded 4-vector per pixel.  The vector L1 (TCP) works in 64-byte chunks; a wave
// gathering 16 B out of 32-byte elements touches 32 chunks per load at 25 % efficiency, and PMC showed the warp
// kernel TA/TCP-bound on exactly that (979 chunk accesses per wave, TA busy 73 %).  16-byte and 8-byte strides
// touch 16 + 8 chunks per tap instead of 32 + 32.  Taps are addressed as uniform base + 32-bit byte offset.
template <typename V2, typename S>
OFX_DEV void bicubic_sample3(const V2 *__restrict__ pa, const S *__restrict__ pb, const BicubicTaps &t, int nx,
                             double &o0, double &o1, double &o2)
{
    const unsigned E = sizeof(S);
    const unsigned r0 = (unsigned) t.row[0] * nx * E, r1 = (unsigned) t.row[1] * nx * E;
    const unsigned r2 = (unsigned) t.row[2] * nx * E, r3 = (unsigned) t.row[3] * nx * E;
    double c0[4], c1[4], c2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned ck = (unsigned) t.col[k] * E;
        const char *ba = reinterpret_cast<const char *>(pa), *bb = reinterpret_cast<const char *>(pb);
        const double2 a0 = ldw2(reinterpret_cast<const V2 *>(ba + 2 * (r0 + ck)));
        const double2 a1 = ldw2(reinterpret_cast<const V2 *>(ba + 2 * (r1 + ck)));
        const double2 a2 = ldw2(reinterpret_cast<const V2 *>(ba + 2 * (r2 + ck)));
        const double2 a3 = ldw2(reinterpret_cast<const V2 *>(ba + 2 * (r3 + ck)));
        const double b0 = ldw(reinterpret_cast<const S *>(bb + r0 + ck));
        const double b1 = ldw(reinterpret_cast<const S *>(bb + r1 + ck));
        const double b2 = ldw(reinterpret_cast<const S *>(bb + r2 + ck));
        const double b3 = ldw(reinterpret_cast<const S *>(bb + r3 + ck));
        c0[k] = cubic_cell(a0.x, a1.x, a2.x, a3.x, t.fy);
        c1[k] = cubic_cell(a0.y, a1.y, a2.y, a3.y, t.fy);
        c2[k] = cubic_cell(b0, b1, b2, b3, t.fy);
    }
    o0 = cubic_cell(c0[0], c0[1], c0[2], c0[3], t.fx);
    o1 = cubic_cell(c1[0], c1[1], c1[2], c1[3], t.fx);
    o2 = cubic_cell(c2[0], c2[1], c2[2], c2[3], t.fx);
}

// ---- stencil point functions ----------------------------------------------------------------------
// Backward-difference divergence at one pixel, src/operators.cpp:35-78.
//   ac = v1[p], al = v1[p-1], bc = v2[p], bu = v2[p-nx].
// Interior columns keep the interior association dxv + (bc - bu) and only drop a v2 term on the
// first / last row (:61-62); the first / last column are written by the reference as (a + b) - c
// (:70-71), a different rounding, so they are separate branches.
OFX_DEV double div_backward(double ac, double al, double bc, double bu, bool lef, bool rig, bool top, bool bot)
{
    if (!lef && !rig) {
        const double dxv = ac - al;
        return top ? dxv + bc : (bot ? dxv - bu : dxv + (bc - bu));
    }
    if (lef) return top ? ac + bc : (bot ? ac - bu : ac + bc - bu);
    return top ? -al + bc : (bot ? -al - bu : -al + bc - bu);
}
