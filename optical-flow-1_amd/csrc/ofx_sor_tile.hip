// ofx_sor_tile.hip -- tolerance-mode SOR sweeps (option sor_exact = 0): K sweeps per launch on LDS tiles.
//
// The exact mode (ofx_sor.hip) keeps the reference's sequential sweep order and is a latency chain; north_star's
// parity bar is an average end-point error below 1e-4 px, and inside that bar the sweeps may be re-ordered.
//
// Horn-Schunck (src/horn_schunck_pyramidal.cpp:31-71,143-231): four colours (i % 2, j % 2) in the order (0,0) (0,1)
// (1,0) (1,1) -- no pixel of a colour reads another pixel of that colour, so a colour step is a plain parallel map and
// the result does not depend on how the image is cut into tiles.  It is the order of oracle.set_sor_order(1), which
// these kernels reproduce bit for bit (same expressions, -ffp-contract=off); against the reference's lexicographic order
// the flows of BASELINE config 3 differ by AEPE 9e-6 (tests/test_gpu_sor.py).
//
// k_hs_tile: one 256-thread workgroup owns a tile of 128 x TH pixels = 64 x TH/2 cells of 2 x 2 pixels (one pixel of
// every colour).  Lane = cell column, wave w owns cell rows w, w + 4, ...; a thread keeps the unknowns and the constant
// operands (A, dif) of its cells in registers for the whole launch and publishes the unknowns in LDS, one plane per
// colour, so that the eight neighbours of a pixel are eight conflict-free 16-byte reads at fixed offsets.  One colour
// step = read neighbours, update, write own plane; a workgroup barrier after every SECOND step (the two colours of a pixel
// row only exchange values inside a wave, see the sweep loop of k_hs_tile).  The dependency cone of one sweep is 4 pixels in x and 2 in
// y (colour (1,1) sees (1,0) sees (0,1) sees (0,0) horizontally, vertically only two links), so K sweeps need a halo of
// 4 K columns and 2 K rows that is recomputed, not exchanged: the workgroup loads the tile once, runs K sweeps and
// stores the inner (128 - 8 K) x (TH - 4 K) pixels.  Sweep s only updates the part of the tile that can still reach
// the output (shrunk by 4 s + 1 / 2 s + 1).  HBM traffic per launch: 40 B read per tile pixel + 16 B written per output
// pixel for K sweeps, against 56 B per pixel and sweep of the one-sweep-per-launch form.
// The reference's replicated border indices are clamped neighbour coordinates; in the tile they are an apron: whoever
// updates a pixel of the image border also writes its mirror image one pixel outside (the halo has room for it).
// Unknowns are ping-pong buffered (neighbouring tiles read this tile's input).  The stopping test looks at the K error
// slots of the previous launch; a loop that ends inside a launch is finished by re-running the first n - k0 sweeps of
// that launch from its untouched input (ofx_loop.h, the scheme of the TV-L1 tile kernel).
#include "ofx_sor_tile.h"
#include "ofx_device.h"
#include "ofx_loop.h"

// between the two colours of a pixel row: the wave's own LDS stores before its own loads (OFX_HST_FOUR_BARRIERS: the A/B build)
#ifdef OFX_HST_FOUR_BARRIERS
#define HST_PAIR_SYNC() __syncthreads()
#else
#define HST_PAIR_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif
#define HST_SOR_W 1.9                // src/horn_schunck_pyramidal.cpp:21
#define HST_W 128                    // tile width in pixels
#define HST_CW 64                    // ... in cells = lanes of a wave
#define HST_PW (HST_CW + 2)          // row pitch of a colour plane in LDS: one spare cell column on either side, and one spare
                                     // cell row above and below, so that EVERY cell can read its neighbours -- the updates of a
                                     // thread's cells are computed unconditionally (straight-line code) and committed by select
// LDS index of cell (row, col) of colour plane `plane`; row in [-1, CR], col in [-1, 64]
template <int CR> OFX_DEV constexpr int hst_at(int plane, int row, int col) { return (plane * (CR + 2) + row + 1) * HST_PW + col + 1; }

template <typename V> OFX_DEV const V *hst_off(const V *base, unsigned byte_off)
{
    return reinterpret_cast<const V *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename V> OFX_DEV V *hst_off(V *base, unsigned byte_off)
{
    return reinterpret_cast<V *>(reinterpret_cast<char *>(base) + byte_off);
}

template <typename T> OFX_DEV double tile_rnd(double x);
template <> OFX_DEV double tile_rnd<double>(double x) { return x; }
template <> OFX_DEV double tile_rnd<float>(double x) { return (double) (float) x; }

// n / d with the reciprocal r = rcp_newton(d) prepared once per launch: quotient estimate, remainder, correction -- the
// compiler's own f64 division (v_div_scale x2, v_rcp, two Newton steps, q = n r, rem = n - d q, v_div_fmas, v_div_fixup) minus
// its scaling and fix-up steps, which are the identity unless an operand or the quotient sits at the edge of the exponent range.
// d = a^2 + alpha^2 lies in [alpha^2, 1e5 + alpha^2] and the numerators are flow-sized, so the result is the IEEE quotient bit
// for bit (tests/test_gpu_sor_tile.py compares whole solves with the oracle's plain divisions).  OFX_HST_IEEE_DIV: the A/B build.
OFX_DEV double hst_div(double n, double d, double r)
{
#ifdef OFX_HST_IEEE_DIV
    (void) r;
    return n / d;
#else
    const double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
#endif
}

// constant operands of a pixel for the whole launch: (I2wx, I2wy), dif and the two reciprocals
struct HstCoef {
    double ax, ay, dif, ru, rv;
};
OFX_DEV HstCoef hst_coef(double2 a, double dif, double alpha2)
{
    HstCoef k;
    k.ax = a.x;
    k.ay = a.y;
    k.dif = dif;
    k.ru = rcp_newton(a.x * a.x + alpha2);
    k.rv = rcp_newton(a.y * a.y + alpha2);
    return k;
}

template <int CR> OFX_DEV void hst_put(double2 *s_u, int li, int lj, double2 v)
{
    // (an image border in the LAST row / column of a tile has its mirror outside the tile: that pixel is never updated here and
    // nothing that reaches the output reads beyond it)
    if (li < 0 || li >= 2 * CR || lj < 0 || lj >= HST_W) return;
    const int plane = ((li & 1) << 1) | (lj & 1);
    s_u[hst_at<CR>(plane, li >> 1, lj >> 1)] = v;
}
// A pixel of the image border also lives one pixel outside the image (clamped neighbour indices, :161-228): in LDS for the
// neighbours in other cells and, when the mirror falls into the pixel's own cell (bottom / right mirror of an even row / column),
// in the registers uc[] of that cell, where its owner reads the in-cell neighbours from.
template <int CR, int CI, int CJ>
OFX_DEV void hst_mirror(double2 *s_u, int li, int lj, int ii, int jj, int nx, int ny, double2 v, double2 (&uc)[4])
{
    const int di = ii == 0 ? -1 : (ii == ny - 1 ? 1 : 0), dj = jj == 0 ? -1 : (jj == nx - 1 ? 1 : 0);
    if (di) {
        hst_put<CR>(s_u, li + di, lj, v);
        if (CI == 0 && di == 1) uc[2 + CJ] = v;
    }
    if (dj) {
        hst_put<CR>(s_u, li, lj + dj, v);
        if (CJ == 0 && dj == 1) uc[CI * 2 + 1] = v;
    }
    if (di && dj) {
        hst_put<CR>(s_u, li + di, lj + dj, v);
        if (CI == 0 && CJ == 0 && di == 1 && dj == 1) uc[3] = v;
    }
}

// neighbour (DI, DJ) of the pixel of colour (CI, CJ) of cell (a_row, lane): from the cell's registers when it is one of the cell's
// own pixels, else from the colour plane it lives in
template <int CR, int CI, int CJ, int DI, int DJ>
OFX_DEV double2 hst_nb(const double2 *s_u, int a_row, int lane, const double2 (&uc)[4])
{
    constexpr int si = CI + DI, sj = CJ + DJ;                             // -1 .. 2
    if constexpr (si >= 0 && si <= 1 && sj >= 0 && sj <= 1) {
        return uc[si * 2 + sj];
    } else {
        constexpr int plane = ((si & 1) << 1) | (sj & 1);
        constexpr int drow = si < 0 ? -1 : si / 2, dcol = sj < 0 ? -1 : sj / 2;
        return s_u[hst_at<CR>(plane, a_row + drow, lane + dcol)];
    }
}

// SOR update of one pixel, src/horn_schunck_pyramidal.cpp:31-71 (the expressions of hs_point_finish in ofx_sor.hip):
// p1..p4 = up-left, up-right, bottom-left, bottom-right, p5..p8 = up, left, bottom, right; the bottom-right corner of the image
// lists its diagonal taps bottom pair first (:222-228).  Returns the new value; e = the squared update (:70).
// CORNER = false: the caller knows the pixel is not the image's bottom-right corner (any row but the last one -- a wave-uniform
// fact), and the operand swap, eight register moves per update, is not even compiled in.
template <typename T, int CR, int CI, int CJ, bool CORNER = true>
OFX_DEV double2 hst_update(const double2 *s_u, int a_row, int lane, const double2 (&uc)[4], const HstCoef &k, double alpha2, bool corner,
                           double &e)
{
    double2 p1 = hst_nb<CR, CI, CJ, -1, -1>(s_u, a_row, lane, uc), p2 = hst_nb<CR, CI, CJ, -1, 1>(s_u, a_row, lane, uc);
    double2 p3 = hst_nb<CR, CI, CJ, 1, -1>(s_u, a_row, lane, uc), p4 = hst_nb<CR, CI, CJ, 1, 1>(s_u, a_row, lane, uc);
    const double2 p5 = hst_nb<CR, CI, CJ, -1, 0>(s_u, a_row, lane, uc), p6 = hst_nb<CR, CI, CJ, 0, -1>(s_u, a_row, lane, uc);
    const double2 p7 = hst_nb<CR, CI, CJ, 1, 0>(s_u, a_row, lane, uc), p8 = hst_nb<CR, CI, CJ, 0, 1>(s_u, a_row, lane, uc);
    if (CORNER && corner) {
        const double2 t1 = p1, t2 = p2;
        p1 = p3; p2 = p4; p3 = t1; p4 = t2;
    }
    const double w = HST_SOR_W;
    // The five coefficients are recomputed in every update (7 of its ~50 instructions).  Left alone the compiler hoists them out
    // of the sweep loop -- 10 more registers per pixel, which spills (and a kernel with a scratch segment starts microseconds
    // later); the empty asm makes the three operands opaque to that motion.
    double ax = k.ax, ay = k.ay, dif = k.dif;
#ifndef OFX_HST_HOIST
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(dif));
#endif
    const double Au = dif * ax, Av = dif * ay;                                // :133-134
    const double Du = ax * ax + alpha2, Dv = ay * ay + alpha2;                // :135-136
    const double D = ax * ay;                                                 // :137
    const double ula = 1. / 12. * (p1.x + p2.x + p3.x + p4.x) + 1. / 6. * (p5.x + p6.x + p7.x + p8.x);
    const double vla = 1. / 12. * (p1.y + p2.y + p3.y + p4.y) + 1. / 6. * (p5.y + p6.y + p7.y + p8.y);
    const double uk = uc[CI * 2 + CJ].x, vk = uc[CI * 2 + CJ].y;
    const double un = tile_rnd<T>((1.0 - w) * uk + hst_div(w * (Au - D * vk + alpha2 * ula), Du, k.ru));   // :66
    const double vn = tile_rnd<T>((1.0 - w) * vk + hst_div(w * (Av - D * un + alpha2 * vla), Dv, k.rv));   // :67
    e = (un - uk) * (un - uk) + (vn - vk) * (vn - vk);                        // :70
    return make_double2(un, vn);
}

// Per-thread pixel flags, one bit per pixel (bit m * 4 + c): kept as three integers in VGPRs and tested bit by bit -- as separate
// booleans they are two SGPRs each, and a thread has up to 24 pixels x 4 flags (the scalar file spilled into the vector one).
struct HstFlags {
    unsigned inside, owner, border;       // inside the image / in the output region and inside / on the image border and inside
};
OFX_DEV bool hst_bit(unsigned f, int b) { return (f >> b) & 1u; }

// one colour of one sweep for the CPT cells of a thread: all updates computed, the active ones committed
template <typename T, int K, int TH, int NW, int CI, int CJ>
OFX_DEV double hst_colour_step(double2 *s_u, int w, int lane, int x0, int y0, int nx, int ny, int s, double2 (&u)[TH / 2 / NW][4],
                               const HstCoef (&kf)[TH / 2 / NW][4], const HstFlags &fl, double alpha2)
{
    constexpr int CR = TH / 2, CPT = CR / NW, C = CI * 2 + CJ;
    const int lj = 2 * lane + CJ;
    // sweep s updates the pixels that can still reach the output: columns [4 s + 1, W - 2 - 4 s], rows [2 s + 1, TH - 2 - 2 s]
    const bool colx = (unsigned) (lj - (4 * s + 1)) <= (unsigned) (HST_W - 3 - 8 * s);
    double e = 0.0;
    bool border = false;
    unsigned f_in = fl.inside, f_own = fl.owner, f_brd = fl.border;
    asm volatile("" : "+v"(f_in), "+v"(f_own), "+v"(f_brd));   // the bit tests stay here (hoisted, each is a pair of SGPRs for the whole loop)
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
        const int li = 2 * a_row + CI;
        const bool rowy = (unsigned) (li - (2 * s + 1)) <= (unsigned) (TH - 3 - 4 * s);          // wave-uniform
        const bool active = colx && rowy && hst_bit(f_in, m * 4 + C);
        double e1;
        double2 un;
        if (y0 + li == ny - 1) un = hst_update<T, CR, CI, CJ, true>(s_u, a_row, lane, u[m], kf[m][C], alpha2, x0 + lj == nx - 1, e1);   // wave-uniform test
        else un = hst_update<T, CR, CI, CJ, false>(s_u, a_row, lane, u[m], kf[m][C], alpha2, false, e1);
        e += (active && hst_bit(f_own, m * 4 + C)) ? e1 : 0.0;
        u[m][C].x = active ? un.x : u[m][C].x;
        u[m][C].y = active ? un.y : u[m][C].y;
        // (an inactive cell must not touch its LDS entry -- outside the image that is somebody's mirror; its store goes to the spare
        // row of its plane and column, a wave-uniform distance away)
        s_u[hst_at<CR>(C, a_row, lane) - (active ? 0 : (a_row + 1) * HST_PW)] = u[m][C];
        border = border || (active && hst_bit(f_brd, m * 4 + C));
    }
    if (border) {                                           // rare: tiles on the image border only
#pragma unroll
        for (int m = 0; m < CPT; m++) {
            const int a_row = w + NW * m;
            const int li = 2 * a_row + CI;
            const bool rowy = (unsigned) (li - (2 * s + 1)) <= (unsigned) (TH - 3 - 4 * s);
            if (colx && rowy && hst_bit(f_brd, m * 4 + C)) hst_mirror<CR, CI, CJ>(s_u, li, lj, y0 + li, x0 + lj, nx, ny, u[m][C], u[m]);
        }
    }
    return e;
}

template <int CR, int CI, int CJ>
OFX_DEV void hst_mirror_init(double2 *s_u, int a_row, int lane, int x0, int y0, int nx, int ny, double2 (&uc)[4])
{
    const int li = 2 * a_row + CI, lj = 2 * lane + CJ, ii = y0 + li, jj = x0 + lj;
    const bool inside = ii >= 0 && ii < ny && jj >= 0 && jj < nx;
    if (inside && (ii == 0 || ii == ny - 1 || jj == 0 || jj == nx - 1)) hst_mirror<CR, CI, CJ>(s_u, li, lj, ii, jj, nx, ny, uc[CI * 2 + CJ], uc);
}

// Launch unit = sweeps [k0, k0 + niter) of every pair of the group (blockIdx.y = pair; bit g of incode: pair g reads U1 and writes
// U0, else the other way; 4 bits of `nit` per pair: its sweep count in this launch).  `check`: the stopping test on the K error
// slots of the previous unit (0 for the re-run of a unit's first sweeps, whose errors all go to the scratch slot slot0).
// NW waves per workgroup: wave w owns the cell rows w, w + NW, ...
template <typename T, int K, int TH, int NW>
__global__ __launch_bounds__(64 * NW) void k_hs_tile(typename Pix<T>::v2 *U0, typename Pix<T>::v2 *U1,
                                                     const typename Pix<T>::v2 *__restrict__ Ag, const T *__restrict__ Difg,
                                                     double *__restrict__ errg, int k0, int check, int slot0, int nx, int ny,
                                                     int tiles_x, double alpha2, double tol, unsigned incode, unsigned runmask,
                                                     unsigned long long nit, int err_stride)
{
    using v2 = typename Pix<T>::v2;
    static_assert((TH / 2) % NW == 0 && TH % 2 == 0 && K >= 1 && K < 16 && HST_W - 8 * K > 0 && TH - 4 * K > 0, "tile geometry");
    constexpr int CR = TH / 2, CPT = CR / NW, HX = 4 * K, HY = 2 * K, OW = HST_W - 2 * HX, OH = TH - 2 * HY;
    extern __shared__ double2 s_u[];                        // [4 colour planes][CR + 2 cell rows][64 + 2 cell columns], hst_at
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    const int niter = (int) ((nit >> (4 * g)) & 15ull);
    const size_t npix = (size_t) nx * ny;
    const bool from1 = (incode >> g) & 1u;
    const v2 *__restrict__ Uin = (from1 ? U1 : U0) + (size_t) g * npix;
    v2 *__restrict__ Uout = (from1 ? U0 : U1) + (size_t) g * npix;
    const v2 *__restrict__ A = Ag + (size_t) g * npix;
    const T *__restrict__ Dif = Difg + (size_t) g * npix;
    double *__restrict__ err = errg + (size_t) g * err_stride;
    double prev[K];
#pragma unroll
    for (int i = 0; i < K; i++) prev[i] = (check && k0 - i > 0) ? loop_fetch_prev(err, k0 - i) : 0.0;

    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * OW - HX, y0 = ty * OH - HY;         // both even: tile parity = image parity
    double2 u[CPT][4];
    HstCoef kf[CPT][4];
    {
        // every load of the thread issued back to back: addresses clamped into the image, the values of pixels outside it zeroed
        double2 la[CPT][4];
        double ld[CPT][4];
#pragma unroll
        for (int m = 0; m < CPT; m++) {
            const int a_row = w + NW * m;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ii = y0 + 2 * a_row + (c >> 1), jj = x0 + 2 * lane + (c & 1);
                const int ic = ii < 0 ? 0 : (ii > ny - 1 ? ny - 1 : ii), jc = jj < 0 ? 0 : (jj > nx - 1 ? nx - 1 : jj);
                // uniform base + 32-bit byte offset (the addressing mode of global_load: one VGPR per address; the host checks
                // that a plane of pairs stays below 4 GiB)
                const unsigned p = ((unsigned) ic * (unsigned) nx + (unsigned) jc) * (unsigned) sizeof(v2);
                u[m][c] = ldw2(hst_off(Uin, p));
                la[m][c] = ldw2(hst_off(A, p));
                ld[m][c] = ldw(hst_off(Dif, p >> 1));
            }
        }
#pragma unroll
        for (int m = 0; m < CPT; m++) {
            const int a_row = w + NW * m;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ii = y0 + 2 * a_row + (c >> 1), jj = x0 + 2 * lane + (c & 1);
                const bool inside = ii >= 0 && ii < ny && jj >= 0 && jj < nx;
                u[m][c].x = inside ? u[m][c].x : 0.0;
                u[m][c].y = inside ? u[m][c].y : 0.0;
                kf[m][c] = hst_coef(la[m][c], ld[m][c], alpha2);
            }
        }
    }
    HstFlags fl = {0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < CPT; m++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int li = 2 * (w + NW * m) + (c >> 1), lj = 2 * lane + (c & 1), ii = y0 + li, jj = x0 + lj;
            const bool inside = ii >= 0 && ii < ny && jj >= 0 && jj < nx;
            const unsigned bit = 1u << (m * 4 + c);
            fl.inside |= inside ? bit : 0u;
            fl.owner |= (inside && li >= HY && li < TH - HY && lj >= HX && lj < HST_W - HX) ? bit : 0u;
            fl.border |= (inside && (ii == 0 || ii == ny - 1 || jj == 0 || jj == nx - 1)) ? bit : 0u;
        }
    }
    asm volatile("" : "+v"(fl.inside), "+v"(fl.owner), "+v"(fl.border));          // keep them integers
    if (check) {                                            // :143 -- the same decision in every wave, before the first barrier
#pragma unroll
        for (int i = 0; i < K; i++)
            if (k0 - i > 0 && !(loop_error_from_sum(wave_allreduce_sum(prev[i]), nx * ny, OFX_CRIT_SQRT_MEAN) > tol)) return;
    }
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
#pragma unroll
        for (int c = 0; c < 4; c++) s_u[hst_at<CR>(c, a_row, lane)] = u[m][c];
    }
    // (the spare ring is read by the discarded updates of the tile's outermost cells only: whatever it holds)
    __syncthreads();                                        // aprons after the planes: a mirror may land in another thread's cell
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
        hst_mirror_init<CR, 0, 0>(s_u, a_row, lane, x0, y0, nx, ny, u[m]);
        hst_mirror_init<CR, 0, 1>(s_u, a_row, lane, x0, y0, nx, ny, u[m]);
        hst_mirror_init<CR, 1, 0>(s_u, a_row, lane, x0, y0, nx, ny, u[m]);
        hst_mirror_init<CR, 1, 1>(s_u, a_row, lane, x0, y0, nx, ny, u[m]);
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < niter; s++) {
        // Two barriers per sweep, not four.  A pixel of colour (0,1) reads the NEW (0,0) values only of its own pixel row -- its
        // own cell and the next cell column, i.e. its own wave (lane = cell column), whose LDS stores and loads are executed in
        // order --; everything it reads from other cell rows, hence other waves, is colour (1,0) / (1,1), which nobody writes
        // before the next barrier (mirrors included: the apron of a border pixel is read by the pixels of that pixel's own row).
        // The same holds for (1,0) -> (1,1).  Fewer, longer phases between barriers: the waves' LDS bursts no longer coincide.
        double e = hst_colour_step<T, K, TH, NW, 0, 0>(s_u, w, lane, x0, y0, nx, ny, s, u, kf, fl, alpha2);
        HST_PAIR_SYNC();
        e += hst_colour_step<T, K, TH, NW, 0, 1>(s_u, w, lane, x0, y0, nx, ny, s, u, kf, fl, alpha2);
        __syncthreads();
        e += hst_colour_step<T, K, TH, NW, 1, 0>(s_u, w, lane, x0, y0, nx, ny, s, u, kf, fl, alpha2);
        HST_PAIR_SYNC();
        e += hst_colour_step<T, K, TH, NW, 1, 1>(s_u, w, lane, x0, y0, nx, ny, s, u, kf, fl, alpha2);
        __syncthreads();
        loop_accumulate(err, slot0 + (check ? s : 0), e, blockIdx.x * NW + w);
    }
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int ii = y0 + 2 * a_row + (c >> 1), jj = x0 + 2 * lane + (c & 1);
            if (hst_bit(fl.owner, m * 4 + c)) stn2(hst_off(Uout, ((unsigned) ii * (unsigned) nx + (unsigned) jj) * (unsigned) sizeof(v2)), u[m][c]);
        }
    }
}

static_assert(OFX_MAX_GROUP * 4 <= 64, "4 bits of `nit` per pair of a lockstep group");
static_assert(OFX_MAX_GROUP <= 32, "one bit of incode / runmask per pair of a lockstep group");

template <typename T, int K, int TH, int NW>
static int hs_tile_launch(ofx_ctx *ctx, int G, typename Pix<T>::v2 *U0, typename Pix<T>::v2 *U1, const typename Pix<T>::v2 *A,
                          const T *Dif, int k0, int check, int slot0, int nx, int ny, double alpha2, double thr, unsigned incode,
                          unsigned runmask, unsigned long long nit, int err_stride)
{
    constexpr int OW = HST_W - 8 * K, OH = TH - 4 * K;
    const int tiles_x = ofx_cdiv(nx, OW), tiles_y = ofx_cdiv(ny, OH);
    const size_t lds = (size_t) 4 * (TH / 2 + 2) * HST_PW * sizeof(double2);
    if (lds > 64 * 1024)
        OFX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hs_tile<T, K, TH, NW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL((k_hs_tile<T, K, TH, NW>), dim3((unsigned) (tiles_x * tiles_y), G), dim3(64 * NW), lds, ctx->stream, U0, U1, A, Dif,
                       ctx->d_err, k0, check, slot0, nx, ny, tiles_x, alpha2, thr, incode, runmask, nit, err_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "hs tile launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}

template <typename T>
int ofx_hs_tile_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *U0, typename Pix<T>::v2 *U1, unsigned *cur,
                      const typename Pix<T>::v2 *A, const T *Dif, int nx, int ny, double alpha2, double TOL, int maxiter, int K,
                      int *niter, double *error, float *ms)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "hs: group of %d pairs", G);
    if ((double) nx * ny * sizeof(typename Pix<T>::v2) >= 4294967296.0)
        return ofx_fail(ctx, OFX_ERR_ARG, "hs: tile sweeps address a level with 32-bit byte offsets (%dx%d is too large)", nx, ny);
    for (int g = 0; g < G; g++) { niter[g] = 0; error[g] = 1000; }                                   // :140
    if (maxiter <= 0 || !(1000.0 > TOL)) return OFX_OK;
    // the result does not depend on K: a lone solve is a chain of launches (fewer, longer ones win: config 3 takes 20.0 / 18.7 ms with
    // K = 2 / 4), a lockstep group is bound by its halo (K = 2: 52 % of the HBM peak on the 56 B unit, K = 4: 43 %)
    if (K <= 0) K = G >= 4 ? 2 : 4;
    if (K > 4) K = 4;
    LoopSpec S;
    S.max_iter = maxiter;
    S.size = nx * ny;
    S.thr = TOL;
    S.crit = OFX_CRIT_SQRT_MEAN;
    S.fixed = ctx->fixed_work != 0;
    S.pairs = K == 2;
    S.fuse = K > 2 ? K : 0;
    S.afac = 0.0;
    if (ctx->chunk > 0) S.chunk = ctx->chunk;
    else {
        // a launch of K sweeps: ~3 us + the tile's 4 K colour steps (~1 us each) per round of 256 workgroups; one poll per ~80 us
        const double tiles = (double) ofx_cdiv(nx, HST_W - 8 * K) * ofx_cdiv(ny, 32) * G;
        const double est_us = 3.0 + 4.0 * K * (tiles < 256.0 ? 1.0 : tiles / 256.0);
        int units = (int) (80.0 / est_us);
        units = units < 1 ? 1 : (units > 8 ? 8 : units);
        S.chunk = units * K;
    }
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    const unsigned b0 = *cur & all;
    // tile geometry (option "sor_tile": 0 = default): 1 = 128 x 32 pixels on 16 waves (one cell row per wave), 2 = 128 x 32 on 8,
    // 3 = 128 x 48 on 12
    const int geom = ctx->sor_tile > 0 ? ctx->sor_tile : 2;
    auto go = [&](int k0, int check, int slot0, double thr, unsigned incode, unsigned runmask, unsigned long long nit) -> int {
#define HST_GO(K_, TH_, NW_)                                                                                                    \
    return hs_tile_launch<T, K_, TH_, NW_>(ctx, G, U0, U1, A, Dif, k0, check, slot0, nx, ny, alpha2, thr, incode, runmask, nit, err_stride)
#define HST_GEOM(K_)                                                                                                            \
    do {                                                                                                                        \
        if (geom == 1) HST_GO(K_, 32, 16);                                                                                      \
        if (geom == 2) HST_GO(K_, 32, 8);                                                                                       \
        HST_GO(K_, 48, 12);                                                                                                     \
    } while (0)
        switch (K) {
        case 1: HST_GEOM(1);
        case 2: HST_GEOM(2);
        case 3: HST_GEOM(3);
        default: HST_GEOM(4);
        }
#undef HST_GEOM
#undef HST_GO
    };
    auto launch = [&](int k, int cnt, double thr) -> int {
        // unit j = k / K reads the buffer pair g started in when j is even
        const unsigned flip = ((k / K) & 1) ? all : 0u;
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) nit |= (unsigned long long) cnt << (4 * g);
        return go(k, 1, k, thr, b0 ^ flip, all, nit);
    };
    auto redo = [&](const int *k_of) -> int {
        unsigned incode = 0, runmask = 0;
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) {
            if (k_of[g] < 0) continue;
            const int n = k_of[g] + 1, j = (n - 1) / K;
            runmask |= 1u << g;
            incode |= (((b0 >> g) ^ (unsigned) j) & 1u) << g;
            nit |= (unsigned long long) (n - j * K) << (4 * g);
        }
        return go(0, 0, S.max_iter, -1.0, incode, runmask, nit);     // no stopping test; errors into the scratch slot
    };
    OFX_TRY(ofx_run_loop_group(ctx, S, G, launch, redo, niter, error, ms));
    unsigned c = 0;
    for (int g = 0; g < G; g++) {
        const unsigned units = (unsigned) ((niter[g] + K - 1) / K);
        c |= (((b0 >> g) ^ units) & 1u) << g;
    }
    *cur = c;
    return OFX_OK;
}

template int ofx_hs_tile_solve<double>(ofx_ctx *, int, double2 *, double2 *, unsigned *, const double2 *, const double *, int, int, double,
                                       double, int, int, int *, double *, float *);
template int ofx_hs_tile_solve<float>(ofx_ctx *, int, float2 *, float2 *, unsigned *, const float2 *, const float *, int, int, double,
                                      double, int, int, int *, double *, float *);

// ============================================================================================================================
// Brox (src/brox_optic_flow_spatial.cpp:129-172,315-390): checkerboard of tiles, the reference's order inside a tile
// ============================================================================================================================
// Red-black sweeps of the 5-point stencil end BASELINE config 4 at AEPE 1.3e-4 from the reference -- over the 1e-4 bar.  The
// solves stop long before they converge (the flow moves by 1.8e-4 when TOL is tightened tenfold), so what a truncated solve
// returns depends on the DIRECTION in which a sweep carries information, and a red-black sweep has none.  What keeps the
// trajectory is to stay lexicographic locally: the image is cut into tiles of 64 rows x TW columns, coloured as a checkerboard;
// a sweep updates the tiles of colour 0, then those of colour 1, each tile in the reference's row-major order, in place -- a
// consistent Gauss-Seidel ordering (every pair of neighbouring pixels has a definite first), so omega = 1.9 stays stable
// (Jacobi coupling across tile borders diverges).  AEPE against the reference at config 4: 1.1e-5 (the reference's own
// 1-vs-16-thread spread is 2.7e-5); oracle order 3 restates the schedule and is reproduced bit for bit.
//
// k_brox_wave: ONE WAVE PER TILE, lane r = row r of the tile's band, marching over the anti-diagonals h = r + j -- lane r
// updates pixel (r, h - r) at step h.  Row-major order only constrains "up and left first", and on an anti-diagonal nobody is
// up or left of anybody, so the steps are the reference's order.  The up neighbour's new value comes from lane r - 1's result of
// the previous step (one wave shift), the left neighbour's from the lane's own; right and down are old values: the lane's own
// next operand and lane r + 1's.  Nothing is read that this launch writes (tiles of the other colour are stable), so the
// operands are prefetched P steps ahead into a register queue and the update is the only dependent chain.
// Band-skewed storage: element (i, j) of a plane lives at band(i) * BS + (i % 64 + j) * 64 + i % 64, so that the 64 pixels of
// an anti-diagonal of a band are 64 consecutive elements -- every access of the wave is one contiguous run.  (Row-major, every
// lane would sit in its own cache line: the layout problem of the exact windows, ofx_sor.hip LaySkew.)  The rows above and
// below the band are other bands: two single-lane loads per step.
#define BRW_SOR_W 1.9                // src/brox_optic_flow_spatial.cpp:25

size_t ofx_band_plane_elems(int nx, int ny) { return (size_t) ((ny + 63) / 64) * (size_t) (nx + 64) * 64; }

// row-major <-> band-skewed copies of one array of a lockstep group (blockIdx.z = pair)
template <typename V, bool IN>
__global__ void k_band(const V *__restrict__ src, V *__restrict__ dst, int nx, int ny, size_t plane)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nx || i >= ny) return;
    const size_t rm = (size_t) blockIdx.z * nx * ny + (size_t) i * nx + j;
    const size_t bd = blockIdx.z * plane + (size_t) (i >> 6) * (size_t) (nx + 64) * 64 + (size_t) ((i & 63) + j) * 64 + (i & 63);
    if (IN) dst[bd] = src[rm];
    else dst[rm] = src[bd];
}
template <typename V, bool IN> int ofx_band_copy(ofx_ctx *ctx, const V *src, V *dst, int nx, int ny, int G)
{
    hipLaunchKernelGGL((k_band<V, IN>), dim3(ofx_cdiv(nx, 64), ofx_cdiv(ny, 4), G), dim3(64, 4), 0, ctx->stream, src, dst, nx, ny,
                       ofx_band_plane_elems(nx, ny));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "band copy launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}
template int ofx_band_copy<double2, true>(ofx_ctx *, const double2 *, double2 *, int, int, int);
template int ofx_band_copy<double2, false>(ofx_ctx *, const double2 *, double2 *, int, int, int);
template int ofx_band_copy<double4, true>(ofx_ctx *, const double4 *, double4 *, int, int, int);
template int ofx_band_copy<double, true>(ofx_ctx *, const double *, double *, int, int, int);
template int ofx_band_copy<float2, true>(ofx_ctx *, const float2 *, float2 *, int, int, int);
template int ofx_band_copy<float2, false>(ofx_ctx *, const float2 *, float2 *, int, int, int);
template int ofx_band_copy<float4, true>(ofx_ctx *, const float4 *, float4 *, int, int, int);
template int ofx_band_copy<float, true>(ofx_ctx *, const float *, float *, int, int, int);

// Buffer loads (ofx_device.h has the stores): a lane whose byte offset lies outside the descriptor's range reads zeros and touches
// no memory, so "this lane has nothing to load" is part of the address -- the instruction is issued unconditionally, in
// straight-line code, and the compiler's s_waitcnt counts stay exact (a load under `if` ends in vmcnt(0) at the join).
OFX_DEV double2 bld2(ofx_rsrc r, unsigned off, const double2 *)
{
    const ofx_u4v w = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    double2 d;
    __builtin_memcpy(&d, &w, 16);
    return d;
}
OFX_DEV double2 bld2(ofx_rsrc r, unsigned off, const float2 *)
{
    const ofx_u2v w = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    float2 f;
    __builtin_memcpy(&f, &w, 8);
    return make_double2((double) f.x, (double) f.y);
}
OFX_DEV double4 bld4(ofx_rsrc r, unsigned off, const double4 *)
{
    // (the builtin's arguments are descriptor, per-lane offset, scalar offset, cache policy: the second half is its own offset,
    // and an out-of-range marker must stay one -- OFX_OOB + 16 would wrap to 0)
    const ofx_u4v a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    const ofx_u4v b = __builtin_amdgcn_raw_buffer_load_b128(r, off == OFX_OOB ? OFX_OOB : off + 16u, 0, 0);
    double2 lo, hi;
    __builtin_memcpy(&lo, &a, 16);
    __builtin_memcpy(&hi, &b, 16);
    return make_double4(lo.x, lo.y, hi.x, hi.y);
}
OFX_DEV double4 bld4(ofx_rsrc r, unsigned off, const float4 *)
{
    const ofx_u4v w = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    float4 f;
    __builtin_memcpy(&f, &w, 16);
    return make_double4((double) f.x, (double) f.y, (double) f.z, (double) f.w);
}
OFX_DEV double bld1(ofx_rsrc r, unsigned off, const double *)
{
    const ofx_u2v w = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    double d;
    __builtin_memcpy(&d, &w, 8);
    return d;
}
OFX_DEV double bld1(ofx_rsrc r, unsigned off, const float *)
{
    const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    float f;
    __builtin_memcpy(&f, &w, 4);
    return (double) f;
}

// operands of one anti-diagonal of a tile, as prefetched: own (du, dv), (Au, Av, Du, Dv), D, psi_s
struct BrwSlot {
    double2 du;
    double4 co;
    double  dm, ps;
};
#define BRW_TW_MAX 128               // columns per tile the edge rows in LDS have room for

// One colour of one sweep (launch k = sweep k; the stopping test of :315 on the error of sweep k - 1).
// The row above the band and the row below it (other bands, stable during the launch) are gathered once into LDS, (du, dv, psi_s)
// per column; in the march every fetch is the same five loads for every lane, so the compiler's wait counters stay exact and the
// queue really is P steps deep.
template <typename T, int P>
__global__ __launch_bounds__(256) void k_brox_wave(typename Pix<T>::v2 *DUg, const typename Pix<T>::v4 *__restrict__ COg,
                                                   const T *__restrict__ Dmg, const T *__restrict__ Psg, double *__restrict__ errg,
                                                   int k, int nx, int ny, int TW, int ntx, int ntiles, int colour, double alpha,
                                                   double tol, unsigned runmask, int err_stride, size_t plane)
{
    using v2 = typename Pix<T>::v2;
    using v4 = typename Pix<T>::v4;
    static_assert(P >= 2, "the right neighbour is the next slot of the queue");
    __shared__ double4 s_edge[4][2][BRW_TW_MAX];                            // [wave][above / below][column of the tile] = (du, dv, psi_s, -)
    const int lane = threadIdx.x & 63, wl = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int wv = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4)) + wl;
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    double *__restrict__ err = errg + (size_t) g * err_stride;
    const double prev = loop_fetch_prev(err, k);
    if (wv >= ntiles) return;
    const int b = wv / ntx, t = wv % ntx;
    if (((b + t) & 1) != colour) return;
    if (!loop_continues(prev, k, nx * ny, tol, OFX_CRIT_SQRT_MEAN)) return;

    const int nbands = (ny + 63) >> 6;
    const unsigned BS = (unsigned) (nx + 64) * 64u;                         // elements per band
    const int nr = ny - b * 64 < 64 ? ny - b * 64 : 64;                     // rows of this band
    const int r = lane, i = b * 64 + r;
    const int c0 = t * TW, c1 = c0 + TW < nx ? c0 + TW : nx;               // columns [c0, c1)
    const size_t boff = (size_t) g * plane + (size_t) b * BS;
    v2 *DU = DUg + boff;
    const v4 *__restrict__ CO = COg + boff;
    const T *__restrict__ Dm = Dmg + boff;
    const T *__restrict__ Ps = Psg + boff;
    const bool top = i == 0, bot = i == ny - 1, rowok = r < nr;
    const bool first = r == 0, last = r == nr - 1;

    // edge rows: pixel (64 b - 1, j) = row 63 of the band above, pixel (64 b + nr, j) = row 0 of the band below
    double4 (*edge)[BRW_TW_MAX] = s_edge[wl];
    for (int jj = c0 + lane; jj < c1; jj += 64) {
        double4 up = make_double4(0.0, 0.0, 0.0, 0.0), dn = up;
        if (b > 0) {
            const unsigned e = (unsigned) (63 + jj) * 64u + 63u;
            const double2 d = ldw2(DU - BS + e);
            up = make_double4(d.x, d.y, ldw(Ps - BS + e), 0.0);
        }
        if (b < nbands - 1) {
            const unsigned e = (unsigned) jj * 64u;
            const double2 d = ldw2(DU + BS + e);
            dn = make_double4(d.x, d.y, ldw(Ps + BS + e), 0.0);
        }
        edge[0][jj - c0] = up;
        edge[1][jj - c0] = dn;
    }
    // the column left of the tile: one (strided) load per lane, before the march
    double2 left_du = make_double2(0.0, 0.0);
    double left_ps = 0.0;
    if (c0 > 0 && rowok) {
        const unsigned e = (unsigned) (r + c0 - 1) * 64u + (unsigned) r;
        left_du = ldw2(DU + e);
        left_ps = ldw(Ps + e);
    }
    const int h0 = c0, hend = c1 - 1 + nr - 1;                              // anti-diagonals of the tile, inclusive
    const int hmax = nx + 63;                                               // last anti-diagonal that is storage
    // An anti-diagonal of the band is 64 pixels, of which only those inside the tile's columns (and one to the right: the right
    // neighbour) are this wave's business -- the others belong to the neighbouring tiles' ramps.  Their lanes are masked out of the
    // loads: unmasked, a tile of TW columns reads TW + 63 + P columns of every row (measured 1.59 x the compulsory bytes,
    // profiles/r04_sor_traffic.json).
    const ofx_rsrc rDU = make_rsrc((void *) DU, BS * (unsigned) sizeof(v2)), rCO = make_rsrc((void *) CO, BS * (unsigned) sizeof(v4));
    const ofx_rsrc rDm = make_rsrc((void *) Dm, BS * (unsigned) sizeof(T)), rPs = make_rsrc((void *) Ps, BS * (unsigned) sizeof(T));
    auto fetch = [&](int hh) -> BrwSlot {
        const unsigned e = (unsigned) (hh < hmax ? hh : hmax) * 64u + (unsigned) r;
        const int jj = hh - r;
        const bool need = rowok && jj >= c0 && jj <= c1;
        BrwSlot s;
        s.du = bld2(rDU, need ? e * (unsigned) sizeof(v2) : OFX_OOB, (const v2 *) nullptr);
        s.co = bld4(rCO, need ? e * (unsigned) sizeof(v4) : OFX_OOB, (const v4 *) nullptr);
        s.dm = bld1(rDm, need ? e * (unsigned) sizeof(T) : OFX_OOB, (const T *) nullptr);
        s.ps = bld1(rPs, need ? e * (unsigned) sizeof(T) : OFX_OOB, (const T *) nullptr);
        return s;
    };
    BrwSlot q[P];
#pragma unroll
    for (int p = 0; p < P; p++) q[p] = fetch(h0 + p);

    const double w = BRW_SOR_W;
    const int tw1 = c1 - c0 - 1;                                            // last column index of the edge rows
    double2 newp = make_double2(0.0, 0.0);                                  // the lane's result of the previous step
    double psp = 0.0;                                                       // ... and its psi_s there
    double e = 0.0;
    for (int hb = h0; hb <= hend; hb += P) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int h = hb + p;
            const BrwSlot s = q[p];
            const BrwSlot &n = q[(p + 1) % P];                              // anti-diagonal h + 1
            const int j = h - r;
            const bool active = rowok && j >= c0 && j < c1 && h <= hend;
            const bool lef = j == 0, rig = j == nx - 1, edge_col = j == c0;
            const double2 c = s.du;
            // lane 0's pixel is in column h, the last lane's in column h - (nr - 1): their rows above / below, from LDS
            const int ju = h - c0, jd = h - (nr - 1) - c0;
            const double4 eu = edge[0][ju < 0 ? 0 : (ju > tw1 ? tw1 : ju)], ed = edge[1][jd < 0 ? 0 : (jd > tw1 ? tw1 : jd)];
            // a missing neighbour is the pixel itself with psi = 0 (:332-388)
            const double sx = wave_shift_down(n.du.x), sy = wave_shift_down(n.du.y), sp = wave_shift_down(n.ps);
            const double ux = wave_shift_up(newp.x), uy = wave_shift_up(newp.y), up_ps = wave_shift_up(psp);
            // (selects on scalars: a ?: on a double2 / double4 goes through a private array, i.e. scratch)
            const double dnx = bot ? c.x : (last ? ed.x : sx), dny = bot ? c.y : (last ? ed.y : sy);
            const double upx = top ? c.x : (first ? eu.x : ux), upy = top ? c.y : (first ? eu.y : uy);
            const double pdn = last ? ed.z : sp, pup = first ? eu.z : up_ps;
            const double rtx = rig ? c.x : n.du.x, rty = rig ? c.y : n.du.y;
            const double lfx = lef ? c.x : (edge_col ? left_du.x : newp.x), lfy = lef ? c.y : (edge_col ? left_du.y : newp.y);
            const double plf = edge_col ? left_ps : psp;
            const double p1 = bot ? 0.0 : 0.5 * (pdn + s.ps);               // src/brox_spatial_mask.cpp:16-93
            const double p2 = top ? 0.0 : 0.5 * (pup + s.ps);
            const double p3 = rig ? 0.0 : 0.5 * (n.ps + s.ps);
            const double p4 = lef ? 0.0 : 0.5 * (plf + s.ps);
            const double ru = rcp_newton(s.co.z), rv = rcp_newton(s.co.w);  // off the dependent chain (hst_div)
            const double div_du = p1 * dnx + p2 * upx + p3 * rtx + p4 * lfx;         // :153-154
            const double div_dv = p1 * dny + p2 * upy + p3 * rty + p4 * lfy;         // :155-156
            const double duk = c.x, dvk = c.y;
            const double dun = tile_rnd<T>((1. - w) * duk + hst_div(w * (s.co.x - s.dm * dvk + alpha * div_du), s.co.z, ru));   // :162
            const double dvn = tile_rnd<T>((1. - w) * dvk + hst_div(w * (s.co.y - s.dm * dun + alpha * div_dv), s.co.w, rv));   // :163
            bst2<false>(rDU, active ? ((unsigned) h * 64u + (unsigned) r) * (unsigned) sizeof(v2) : OFX_OOB, make_double2(dun, dvn), (const v2 *) nullptr);
            e += active ? (dun - duk) * (dun - duk) + (dvn - dvk) * (dvn - dvk) : 0.0;     // :166
            newp = make_double2(dun, dvn);
            psp = s.ps;
            q[p] = fetch(h + P);
        }
    }
    loop_accumulate(err, k, e, wv);
}

template <typename T>
int ofx_brox_wave_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *DUb, const typename Pix<T>::v4 *COb, const T *Dmb, const T *Psb,
                        int nx, int ny, double alpha, double TOL, int maxiter, int *niter, double *error, float *ms)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "brox: group of %d pairs", G);
    const size_t plane = ofx_band_plane_elems(nx, ny);
    if ((double) plane * sizeof(typename Pix<T>::v4) >= 4294967296.0)
        return ofx_fail(ctx, OFX_ERR_ARG, "brox: tile sweeps address a level with 32-bit offsets (%dx%d is too large)", nx, ny);
    for (int g = 0; g < G; g++) { niter[g] = 0; error[g] = 1000; }                                    // :312
    if (maxiter <= 0 || !(1000.0 > TOL)) return OFX_OK;
    const int TW = ctx->sor_tile_w > 0 ? (ctx->sor_tile_w > BRW_TW_MAX ? BRW_TW_MAX : ctx->sor_tile_w) : BRW_TW_MAX;
    const int ntx = ofx_cdiv(nx, TW), ntiles = ntx * ((ny + 63) / 64);
    LoopSpec S;
    S.max_iter = maxiter;
    S.size = nx * ny;
    S.thr = TOL;
    S.crit = OFX_CRIT_SQRT_MEAN;
    S.fixed = ctx->fixed_work != 0;
    S.pairs = false;
    if (ctx->chunk > 0) S.chunk = ctx->chunk;
    else {
        // a sweep = two launches of TW + 63 dependent steps (~0.3 us each)
        const double est_us = 2.0 * (TW + 63) * 0.3;
        int c = (int) (120.0 / est_us);
        S.chunk = c < 2 ? 2 : (c > 16 ? 16 : c);
    }
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    const dim3 grid((unsigned) ofx_cdiv(ntiles, 4), G), block(256);
    const int P = ctx->sor_wave_p > 0 ? ctx->sor_wave_p : 4;                // steps of prefetch (option "sor_wave_p": 2 | 4 | 6 | 8)
    auto launch = [&](int k, int, double thr) -> int {
        for (int col = 0; col < 2; col++) {
#define BRW_GO(P_)                                                                                                             \
    hipLaunchKernelGGL((k_brox_wave<T, P_>), grid, block, 0, ctx->stream, DUb, COb, Dmb, Psb, ctx->d_err, k, nx, ny, TW, ntx, ntiles, \
                       col, alpha, thr, all, err_stride, plane)
            if (P <= 2) BRW_GO(2);
            else if (P <= 4) BRW_GO(4);
            else if (P <= 6) BRW_GO(6);
            else BRW_GO(8);
#undef BRW_GO
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "brox wave launch failed: %s", hipGetErrorString(e));
        return OFX_OK;
    };
    return ofx_run_loop_group(ctx, S, G, launch, [](const int *) { return OFX_OK; }, niter, error, ms);
}
template int ofx_brox_wave_solve<double>(ofx_ctx *, int, double2 *, const double4 *, const double *, const double *, int, int, double,
                                         double, int, int *, double *, float *);
template int ofx_brox_wave_solve<float>(ofx_ctx *, int, float2 *, const float4 *, const float *, const float *, int, int, double, double,
                                        int, int *, double *, float *);

// ============================================================================================================================
// Brox, the coarser levels: red-black sweeps, K per launch on LDS tiles (k_brox_tile)
// ============================================================================================================================
// Below the levels that k_brox_wave sweeps, the order is the red-black one of k_brox_sor (ofx_sor.hip; (i + j) even first) -- the
// result does not depend on tiles, K or the group -- but as ONE launch per K sweeps instead of two per sweep: k_brox_sor touches
// every cache line in both colour passes (190 B per pixel-sweep at the HBM side, 2.4 x the 80 B unit, profiles/r04_sor_traffic.json)
// and a lone solve on these levels is a chain of ~8 us launches.  The tile scheme is k_hs_tile's: cells of 2 x 2 pixels (two of
// either colour), lane = cell column, the unknowns AND psi_s published in LDS colour planes, a pixel's other operands -- (Au, Av,
// Du, Dv), D -- in registers; the 5-point stencil needs two LDS reads per operand plane (the other two neighbours are the cell's
// own pixels), a sweep's dependency cone is 2 pixels in x and y, so K sweeps recompute a halo of 2 K.  A missing neighbour at the
// image border is the pixel itself with psi = 0 (:332-388), i.e. a select, no apron.
#define BRT_PS_BASE(CR) (4 * ((CR) + 2) * HST_PW)     // the psi_s planes follow the four planes of unknowns (as doubles)

template <int CR, int CI, int CJ, int DI, int DJ>
OFX_DEV double2 brt_nb_u(const double2 *s_u, int a_row, int lane, const double2 (&uc)[4])
{
    constexpr int si = CI + DI, sj = CJ + DJ;
    if constexpr (si >= 0 && si <= 1 && sj >= 0 && sj <= 1) {
        return uc[si * 2 + sj];
    } else {
        constexpr int plane = ((si & 1) << 1) | (sj & 1);
        constexpr int drow = si < 0 ? -1 : si / 2, dcol = sj < 0 ? -1 : sj / 2;
        return s_u[hst_at<CR>(plane, a_row + drow, lane + dcol)];
    }
}
template <int CR, int CI, int CJ, int DI, int DJ>
OFX_DEV double brt_nb_ps(const double *s_ps, int a_row, int lane, const double (&pc)[4])
{
    constexpr int si = CI + DI, sj = CJ + DJ;
    if constexpr (si >= 0 && si <= 1 && sj >= 0 && sj <= 1) {
        return pc[si * 2 + sj];
    } else {
        constexpr int plane = ((si & 1) << 1) | (sj & 1);
        constexpr int drow = si < 0 ? -1 : si / 2, dcol = sj < 0 ? -1 : sj / 2;
        return s_ps[hst_at<CR>(plane, a_row + drow, lane + dcol)];
    }
}

struct BrtCoef {
    double au, av, du, dv, dm;
};

// SOR update of one pixel, src/brox_optic_flow_spatial.cpp:129-172 (the expressions of brox_point_finish in ofx_sor.hip)
template <typename T, int CR, int CI, int CJ>
OFX_DEV double2 brt_update(const double2 *s_u, const double *s_ps, int a_row, int lane, const double2 (&uc)[4], const double (&pc)[4],
                           const BrtCoef &k, double alpha, bool top, bool bot, bool lef, bool rig, double &e)
{
    const double2 c = uc[CI * 2 + CJ];
    const double psc = pc[CI * 2 + CJ];
    const double2 dn0 = brt_nb_u<CR, CI, CJ, 1, 0>(s_u, a_row, lane, uc), up0 = brt_nb_u<CR, CI, CJ, -1, 0>(s_u, a_row, lane, uc);
    const double2 rt0 = brt_nb_u<CR, CI, CJ, 0, 1>(s_u, a_row, lane, uc), lf0 = brt_nb_u<CR, CI, CJ, 0, -1>(s_u, a_row, lane, uc);
    const double pdn = brt_nb_ps<CR, CI, CJ, 1, 0>(s_ps, a_row, lane, pc), pup = brt_nb_ps<CR, CI, CJ, -1, 0>(s_ps, a_row, lane, pc);
    const double prt = brt_nb_ps<CR, CI, CJ, 0, 1>(s_ps, a_row, lane, pc), plf = brt_nb_ps<CR, CI, CJ, 0, -1>(s_ps, a_row, lane, pc);
    const double dnx = bot ? c.x : dn0.x, dny = bot ? c.y : dn0.y, upx = top ? c.x : up0.x, upy = top ? c.y : up0.y;
    const double rtx = rig ? c.x : rt0.x, rty = rig ? c.y : rt0.y, lfx = lef ? c.x : lf0.x, lfy = lef ? c.y : lf0.y;
    const double p1 = bot ? 0.0 : 0.5 * (pdn + psc);                    // src/brox_spatial_mask.cpp:16-93
    const double p2 = top ? 0.0 : 0.5 * (pup + psc);
    const double p3 = rig ? 0.0 : 0.5 * (prt + psc);
    const double p4 = lef ? 0.0 : 0.5 * (plf + psc);
    const double w = BRW_SOR_W;
    double du = k.du, dv = k.dv;
#ifndef OFX_HST_HOIST
    asm volatile("" : "+v"(du), "+v"(dv));                              // the reciprocals stay in the loop (registers, k_hs_tile)
#endif
    const double ru = rcp_newton(du), rv = rcp_newton(dv);
    const double div_du = p1 * dnx + p2 * upx + p3 * rtx + p4 * lfx;    // :153-154
    const double div_dv = p1 * dny + p2 * upy + p3 * rty + p4 * lfy;    // :155-156
    const double duk = c.x, dvk = c.y;
    const double dun = tile_rnd<T>((1. - w) * duk + hst_div(w * (k.au - k.dm * dvk + alpha * div_du), du, ru));   // :162
    const double dvn = tile_rnd<T>((1. - w) * dvk + hst_div(w * (k.av - k.dm * dun + alpha * div_dv), dv, rv));   // :163
    e = (dun - duk) * (dun - duk) + (dvn - dvk) * (dvn - dvk);          // :166
    return make_double2(dun, dvn);
}

// one pixel colour (CI, CJ) of the cells of a thread in sweep s (red = (0,0) and (1,1), black = (0,1) and (1,0))
template <typename T, int K, int TH, int NW, int CI, int CJ>
OFX_DEV double brt_colour_step(double2 *s_u, const double *s_ps, int w, int lane, int x0, int y0, int nx, int ny, int s,
                               double2 (&u)[TH / 2 / NW][4], const double (&ps)[TH / 2 / NW][4], const BrtCoef (&kf)[TH / 2 / NW][4],
                               const HstFlags &fl, double alpha)
{
    constexpr int CR = TH / 2, CPT = CR / NW, C = CI * 2 + CJ;
    const int lj = 2 * lane + CJ, jj = x0 + lj;
    const bool colx = (unsigned) (lj - (2 * s + 1)) <= (unsigned) (HST_W - 3 - 4 * s);      // columns [2 s + 1, W - 2 - 2 s]
    const bool lef = jj == 0, rig = jj == nx - 1;
    unsigned f_in = fl.inside, f_own = fl.owner;
    asm volatile("" : "+v"(f_in), "+v"(f_own));
    double e = 0.0;
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
        const int li = 2 * a_row + CI, ii = y0 + li;
        const bool rowy = (unsigned) (li - (2 * s + 1)) <= (unsigned) (TH - 3 - 4 * s);      // rows [2 s + 1, TH - 2 - 2 s]
        const bool active = colx && rowy && hst_bit(f_in, m * 4 + C);
        double e1;
        const double2 un = brt_update<T, CR, CI, CJ>(s_u, s_ps, a_row, lane, u[m], ps[m], kf[m][C], alpha, ii == 0, ii == ny - 1, lef, rig, e1);
        e += (active && hst_bit(f_own, m * 4 + C)) ? e1 : 0.0;
        u[m][C].x = active ? un.x : u[m][C].x;
        u[m][C].y = active ? un.y : u[m][C].y;
        s_u[hst_at<CR>(C, a_row, lane)] = u[m][C];                       // (an inactive cell rewrites its own value: no aprons here)
    }
    return e;
}

template <typename T, int K, int TH, int NW>
__global__ __launch_bounds__(64 * NW) void k_brox_tile(typename Pix<T>::v2 *DU0, typename Pix<T>::v2 *DU1,
                                                       const typename Pix<T>::v4 *__restrict__ COg, const T *__restrict__ Dmg,
                                                       const T *__restrict__ Psg, double *__restrict__ errg, int k0, int check, int slot0,
                                                       int nx, int ny, int tiles_x, double alpha, double tol, unsigned incode,
                                                       unsigned runmask, unsigned long long nit, int err_stride)
{
    using v2 = typename Pix<T>::v2;
    using v4 = typename Pix<T>::v4;
    static_assert((TH / 2) % NW == 0 && TH % 2 == 0 && K >= 1 && K < 16 && HST_W - 4 * K > 0 && TH - 4 * K > 0, "tile geometry");
    constexpr int CR = TH / 2, CPT = CR / NW, HX = 2 * K, HY = 2 * K, OW = HST_W - 2 * HX, OH = TH - 2 * HY;
    extern __shared__ double2 s_u[];                        // four planes of unknowns, then four of psi_s (hst_at, BRT_PS_BASE)
    double *s_ps = reinterpret_cast<double *>(s_u + BRT_PS_BASE(CR));
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int g = blockIdx.y;
    if (!((runmask >> g) & 1u)) return;
    const int niter = (int) ((nit >> (4 * g)) & 15ull);
    const size_t npix = (size_t) nx * ny;
    const bool from1 = (incode >> g) & 1u;
    const v2 *__restrict__ Uin = (from1 ? DU1 : DU0) + (size_t) g * npix;
    v2 *__restrict__ Uout = (from1 ? DU0 : DU1) + (size_t) g * npix;
    const v4 *__restrict__ CO = COg + (size_t) g * npix;
    const T *__restrict__ Dm = Dmg + (size_t) g * npix;
    const T *__restrict__ Ps = Psg + (size_t) g * npix;
    double *__restrict__ err = errg + (size_t) g * err_stride;
    double prev[K];
#pragma unroll
    for (int i = 0; i < K; i++) prev[i] = (check && k0 - i > 0) ? loop_fetch_prev(err, k0 - i) : 0.0;

    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * OW - HX, y0 = ty * OH - HY;         // both even: tile parity = image parity
    double2 u[CPT][4];
    double ps[CPT][4];
    BrtCoef kf[CPT][4];
    HstFlags fl = {0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int li = 2 * a_row + (c >> 1), lj = 2 * lane + (c & 1), ii = y0 + li, jj = x0 + lj;
            const int ic = ii < 0 ? 0 : (ii > ny - 1 ? ny - 1 : ii), jc = jj < 0 ? 0 : (jj > nx - 1 ? nx - 1 : jj);
            const unsigned p = (unsigned) ic * (unsigned) nx + (unsigned) jc;                       // element index; clamped into the image
            u[m][c] = ldw2(hst_off(Uin, p * (unsigned) sizeof(v2)));
            const double4 co = ldw4(hst_off(CO, p * (unsigned) sizeof(v4)));
            kf[m][c].au = co.x;
            kf[m][c].av = co.y;
            kf[m][c].du = co.z;
            kf[m][c].dv = co.w;
            kf[m][c].dm = ldw(hst_off(Dm, p * (unsigned) sizeof(T)));
            ps[m][c] = ldw(hst_off(Ps, p * (unsigned) sizeof(T)));
            const bool inside = ii >= 0 && ii < ny && jj >= 0 && jj < nx;
            const unsigned bit = 1u << (m * 4 + c);
            fl.inside |= inside ? bit : 0u;
            fl.owner |= (inside && li >= HY && li < TH - HY && lj >= HX && lj < HST_W - HX) ? bit : 0u;
        }
    }
    asm volatile("" : "+v"(fl.inside), "+v"(fl.owner));
    if (check) {                                            // :315 -- the same decision in every wave, before the first barrier
#pragma unroll
        for (int i = 0; i < K; i++)
            if (k0 - i > 0 && !(loop_error_from_sum(wave_allreduce_sum(prev[i]), nx * ny, OFX_CRIT_SQRT_MEAN) > tol)) return;
    }
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s_u[hst_at<CR>(c, a_row, lane)] = u[m][c];
            s_ps[hst_at<CR>(c, a_row, lane)] = ps[m][c];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < niter; s++) {
        double e = brt_colour_step<T, K, TH, NW, 0, 0>(s_u, s_ps, w, lane, x0, y0, nx, ny, s, u, ps, kf, fl, alpha);
        e += brt_colour_step<T, K, TH, NW, 1, 1>(s_u, s_ps, w, lane, x0, y0, nx, ny, s, u, ps, kf, fl, alpha);
        __syncthreads();
        e += brt_colour_step<T, K, TH, NW, 0, 1>(s_u, s_ps, w, lane, x0, y0, nx, ny, s, u, ps, kf, fl, alpha);
        e += brt_colour_step<T, K, TH, NW, 1, 0>(s_u, s_ps, w, lane, x0, y0, nx, ny, s, u, ps, kf, fl, alpha);
        __syncthreads();
        loop_accumulate(err, slot0 + (check ? s : 0), e, blockIdx.x * NW + w);
    }
#pragma unroll
    for (int m = 0; m < CPT; m++) {
        const int a_row = w + NW * m;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int ii = y0 + 2 * a_row + (c >> 1), jj = x0 + 2 * lane + (c & 1);
            if (hst_bit(fl.owner, m * 4 + c)) stn2(hst_off(Uout, ((unsigned) ii * (unsigned) nx + (unsigned) jj) * (unsigned) sizeof(v2)), u[m][c]);
        }
    }
}

// pairs whose bit of `mask` is set: dst <- src (the tile sweeps' result back into the buffer the rest of the level reads)
template <typename V> __global__ void k_copy_pairs(const V *__restrict__ src, V *__restrict__ dst, size_t npix, unsigned mask)
{
    if (!((mask >> blockIdx.y) & 1u)) return;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npix) dst[blockIdx.y * npix + i] = src[blockIdx.y * npix + i];
}

template <typename T, int K, int TH, int NW>
static int brox_tile_launch(ofx_ctx *ctx, int G, typename Pix<T>::v2 *DU0, typename Pix<T>::v2 *DU1, const typename Pix<T>::v4 *CO,
                            const T *Dm, const T *Ps, int k0, int check, int slot0, int nx, int ny, double alpha, double thr,
                            unsigned incode, unsigned runmask, unsigned long long nit, int err_stride)
{
    constexpr int OW = HST_W - 4 * K, OH = TH - 4 * K;
    const int tiles_x = ofx_cdiv(nx, OW), tiles_y = ofx_cdiv(ny, OH);
    const size_t lds = (size_t) BRT_PS_BASE(TH / 2) * (sizeof(double2) + sizeof(double));
    if (lds > 64 * 1024)
        OFX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_brox_tile<T, K, TH, NW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL((k_brox_tile<T, K, TH, NW>), dim3((unsigned) (tiles_x * tiles_y), G), dim3(64 * NW), lds, ctx->stream, DU0, DU1, CO,
                       Dm, Ps, ctx->d_err, k0, check, slot0, nx, ny, tiles_x, alpha, thr, incode, runmask, nit, err_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "brox tile launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}

// One solve of the coarser levels; DU0 holds the incoming (du, dv), DU1 is the second buffer of the ping-pong; on return the
// result is in DU0 again for every pair (k_copy_pairs for those whose last launch wrote DU1).
template <typename T>
int ofx_brox_tile_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *DU0, typename Pix<T>::v2 *DU1, const typename Pix<T>::v4 *CO,
                        const T *Dm, const T *Ps, int nx, int ny, double alpha, double TOL, int maxiter, int K, int *niter, double *error,
                        float *ms)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "brox: group of %d pairs", G);
    if ((double) nx * ny * sizeof(typename Pix<T>::v4) >= 4294967296.0)
        return ofx_fail(ctx, OFX_ERR_ARG, "brox: tile sweeps address a level with 32-bit byte offsets (%dx%d is too large)", nx, ny);
    for (int g = 0; g < G; g++) { niter[g] = 0; error[g] = 1000; }                                   // :312
    if (maxiter <= 0 || !(1000.0 > TOL)) return OFX_OK;
    if (K <= 0) K = 4;                                      // measured (config 4): one pair 42.6 (k_brox_sor) / 45.5 / 40.6 / 37.8 ms for K = 1 / 2 / 4,
                                                            // batches of 48 37.4 / 41.6 / 45.3 / 45.3 % of the HBM peak on the 80 B unit
    if (K > 4) K = 4;
    if (K == 3) K = 2;                                      // instantiated: 1, 2, 4
    LoopSpec S;
    S.max_iter = maxiter;
    S.size = nx * ny;
    S.thr = TOL;
    S.crit = OFX_CRIT_SQRT_MEAN;
    S.fixed = ctx->fixed_work != 0;
    S.pairs = K == 2;
    S.fuse = K > 2 ? K : 0;
    S.afac = 0.0;
    if (ctx->chunk > 0) S.chunk = ctx->chunk;
    else {
        const double tiles = (double) ofx_cdiv(nx, HST_W - 4 * K) * ofx_cdiv(ny, 32 - 4 * K) * G;
        const double est_us = 3.0 + 2.5 * K * (tiles < 256.0 ? 1.0 : tiles / 256.0);
        int units = (int) (80.0 / est_us);
        units = units < 1 ? 1 : (units > 8 ? 8 : units);
        S.chunk = units * K;
    }
    const int err_stride = (S.max_iter + 1) * OFX_NSHARD;
    const unsigned all = (G >= 32) ? 0xFFFFFFFFu : ((1u << G) - 1u);
    auto go = [&](int k0, int check, int slot0, double thr, unsigned incode, unsigned runmask, unsigned long long nit) -> int {
        switch (K) {
        case 1: return brox_tile_launch<T, 1, 32, 16>(ctx, G, DU0, DU1, CO, Dm, Ps, k0, check, slot0, nx, ny, alpha, thr, incode, runmask, nit, err_stride);
        case 2: return brox_tile_launch<T, 2, 32, 16>(ctx, G, DU0, DU1, CO, Dm, Ps, k0, check, slot0, nx, ny, alpha, thr, incode, runmask, nit, err_stride);
        default: return brox_tile_launch<T, 4, 32, 16>(ctx, G, DU0, DU1, CO, Dm, Ps, k0, check, slot0, nx, ny, alpha, thr, incode, runmask, nit, err_stride);
        }
    };
    auto launch = [&](int k, int cnt, double thr) -> int {
        const unsigned flip = ((k / K) & 1) ? all : 0u;      // unit j reads DU0 when j is even (every pair starts in DU0)
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) nit |= (unsigned long long) cnt << (4 * g);
        return go(k, 1, k, thr, flip, all, nit);
    };
    auto redo = [&](const int *k_of) -> int {
        unsigned incode = 0, runmask = 0;
        unsigned long long nit = 0;
        for (int g = 0; g < G; g++) {
            if (k_of[g] < 0) continue;
            const int n = k_of[g] + 1, j = (n - 1) / K;
            runmask |= 1u << g;
            incode |= ((unsigned) j & 1u) << g;
            nit |= (unsigned long long) (n - j * K) << (4 * g);
        }
        return go(0, 0, S.max_iter, -1.0, incode, runmask, nit);
    };
    OFX_TRY(ofx_run_loop_group(ctx, S, G, launch, redo, niter, error, ms));
    unsigned in1 = 0;
    for (int g = 0; g < G; g++) in1 |= ((unsigned) ((niter[g] + K - 1) / K) & 1u) << g;
    if (in1) {
        const size_t npix = (size_t) nx * ny;
        hipLaunchKernelGGL((k_copy_pairs<typename Pix<T>::v2>), dim3((unsigned) ((npix + 255) / 256), G), dim3(256), 0, ctx->stream,
                           (const typename Pix<T>::v2 *) DU1, DU0, npix, in1);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "copy launch failed: %s", hipGetErrorString(e));
    }
    return OFX_OK;
}
template int ofx_brox_tile_solve<double>(ofx_ctx *, int, double2 *, double2 *, const double4 *, const double *, const double *, int, int,
                                         double, double, int, int, int *, double *, float *);
template int ofx_brox_tile_solve<float>(ofx_ctx *, int, float2 *, float2 *, const float4 *, const float *, const float *, int, int, double,
                                        double, int, int, int *, double *, float *);
