// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// extern "C" trampolines over the *unmodified* reference library, so that the
// reference's own compiled code can be driven through ctypes from tests/ and
// from bench.py's cpu_baseline leg.  This file is our own code; it contains no
// reference source.  It is compiled together with the reference's L1-L3
// sources *where they lie* under /root/reference/src by oracle/Makefile into
// oracle/_ref/libofref.so (git-ignored, travels to the GPU box as a binary).
//
// The reference library has C++ linkage and `bool`/default arguments
// (tvl1flow.h:36-70, operators.h:29-134, bicubic_interpolation.h:16-52,
// zoom.h:20-63, utils.h:27-32,119); the trampolines only flatten that to a
// C ABI.  ofpix_t is `double` (of.h:4-10).
#include <stdexcept>

#include "tvl1flow.h"
#include "operators.h"
#include "bicubic_interpolation.h"
#include "zoom.h"
#include "utils.h"
#include "horn_schunck.h"
#include "brox_optic_flow.h"
#include "brox_spatial_mask.h"
#include "tvl1occflow_solvers.h"
#include "tvl1occflow_tv_rof_box.h"

#include <cstdlib>
#include <new>

#ifdef _OPENMP
#include <omp.h>
#endif

// Array allocations of THIS library return zeroed memory (the library is linked -Bsymbolic, nothing outside it is
// affected).  Needed for exactly one function: Solver_wrt_chi (tvl1occflow_solvers.cpp:239-263) reads its dual variable
// eta straight after `new[]` -- the source says so itself ("#warning eta1 and eta2 are used uninitialized") -- so its
// result depends on the heap's history.  With zero-filled arrays the function is deterministic and can pin the
// restatement, which takes eta as explicit, zero-initialised state.  Every other function writes its arrays before
// reading them.
void *operator new[](std::size_t n)
{
    void *p = std::calloc(1, n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

extern "C" {

int ref_sizeof_pix(void) { return (int) sizeof(ofpix_t); }

void ref_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void) n;
#endif
}

int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ref_divergence(const double *v1, const double *v2, double *div, int nx, int ny)
{ divergence(v1, v2, div, nx, ny); }

void ref_forward_gradient(const double *f, double *fx, double *fy, int nx, int ny)
{ forward_gradient(f, fx, fy, nx, ny); }

void ref_centered_gradient(const double *f, double *dx, double *dy, int nx, int ny)
{ centered_gradient(f, dx, dy, nx, ny, 1); }

void ref_Dxx(const double *I, double *o, int nx, int ny) { Dxx(I, o, nx, ny, 1); }
void ref_Dyy(const double *I, double *o, int nx, int ny) { Dyy(I, o, nx, ny, 1); }
void ref_Dxy(const double *I, double *o, int nx, int ny) { Dxy(I, o, nx, ny, 1); }

// returns 1 when the reference throws (operators.cpp:520-522)
int ref_gaussian(double *I, int nx, int ny, double sigma)
{
    try { gaussian(I, nx, ny, sigma); }
    catch (const std::runtime_error &) { return 1; }
    return 0;
}

double ref_bicubic_at(const double *in, double uu, double vv, int nx, int ny, int border_out)
{ return bicubic_interpolation_at(in, uu, vv, nx, ny, border_out != 0); }

void ref_bicubic_warp(const double *in, const double *u, const double *v, double *out,
                      int nx, int ny, int border_out)
{ bicubic_interpolation_warp(in, u, v, out, nx, ny, border_out != 0); }

void ref_zoom_size(int nx, int ny, int *nxx, int *nyy, double factor)
{ zoom_size(nx, ny, nxx, nyy, factor); }

int ref_zoom_out(const double *I, double *Iout, int nx, int ny, double factor)
{
    try { zoom_out(I, Iout, nx, ny, factor); }
    catch (const std::runtime_error &) { return 1; }
    return 0;
}

void ref_zoom_in(const double *I, double *Iout, int nx, int ny, int nxx, int nyy)
{ zoom_in(I, Iout, nx, ny, nxx, nyy); }

void ref_image_normalization_2(const double *I1, const double *I2, double *I1n, double *I2n, int size)
{ image_normalization_2(I1, I2, I1n, I2n, size); }

void ref_tvl1_single_scale(double *I0, double *I1, double *u1, double *u2, int nx, int ny,
                           double tau, double lambda, double theta, int warps,
                           double epsilon, int verbose)
{ Dual_TVL1_optic_flow(I0, I1, u1, u2, nx, ny, tau, lambda, theta, warps, epsilon, verbose != 0); }

int ref_tvl1_multiscale(double *I0, double *I1, double *u1, double *u2, int nx, int ny,
                        double tau, double lambda, double theta, int nscales, double zfactor,
                        int warps, double epsilon, int verbose)
{
    try {
        Dual_TVL1_optic_flow_multiscale(I0, I1, u1, u2, nx, ny, tau, lambda, theta,
                                        nscales, zfactor, warps, epsilon, verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    return 0;
}

void ref_hs_single_scale(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                         double alpha, int warps, double TOL, int maxiter, int verbose)
{ horn_schunck_optical_flow(I1, I2, u, v, nx, ny, alpha, warps, TOL, maxiter, verbose != 0); }

int ref_hs_pyramidal(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                     double alpha, int nscales, double zfactor, int warps, double TOL,
                     int maxiter, int verbose)
{
    try {
        horn_schunck_pyramidal(I1, I2, u, v, nx, ny, alpha, nscales, zfactor, warps, TOL,
                               maxiter, verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    return 0;
}

int ref_brox_spatial(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                     double alpha, double gamma, int nscales, double nu, double TOL,
                     int inner_iter, int outer_iter, int verbose)
{
    try {
        brox_optic_flow_spatial(I1, I2, u, v, nx, ny, alpha, gamma, nscales, nu, TOL,
                                inner_iter, outer_iter, verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    return 0;
}

int ref_brox_temporal(const double *I, double *u, double *v, int nx, int ny, int frames, double alpha,
                      double gamma, int nscales, double nu, double TOL, int inner_iter, int outer_iter, int verbose)
{
    if (frames <= 2) return 2;
    try {
        brox_optic_flow_temporal(I, u, v, nx, ny, frames, alpha, gamma, nscales, nu, TOL, inner_iter, outer_iter,
                                 verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    return 0;
}

void ref_centered_gradient3(const double *in, double *dx, double *dy, double *dz, int nx, int ny, int nz)
{ centered_gradient3(in, dx, dy, dz, nx, ny, nz); }

void ref_image_normalization_1(const double *I, double *In, int size) { image_normalization_1(I, In, size); }

double ref_bicubic_at_color(const double *in, double uu, double vv, int nx, int ny, int nz, int k, int border_out)
{ return bicubic_interpolation_at_color(in, uu, vv, nx, ny, nz, k, border_out != 0); }

void ref_getminmax(double *mn, double *mx, const double *x, int n) { getminmax(mn, mx, x, n); }

void ref_hs_classic(double *u, double *v, const double *a, const double *b, int w, int h, int n, double alpha)
{ hs(u, v, const_cast<double *>(a), const_cast<double *>(b), w, h, n, alpha); }


// ---- SURVEY 8(f)4 colour operators -------------------------------------------------------------------------------
void ref_bicubic_warp_color(const double *in, const double *u, const double *v, double *out, int nx, int ny, int nz,
                            int border_out)
{ bicubic_interpolation_warp_color(in, u, v, out, nx, ny, nz, border_out != 0); }

void ref_image_normalization_2_color(const double *I1, const double *I2, double *I1n, double *I2n, int size, int nz)
{ image_normalization_2_color(I1, I2, I1n, I2n, size, nz); }

void ref_image_normalization_3(double *I0, double *I1, double *I2, int size) { image_normalization_3(I0, I1, I2, size); }

void ref_image_normalization_4(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *I_1n,
                               double *I0n, double *I1n, double *filtI0n, int size)
{ image_normalization_4(I_1, I0, I1, filtI0, I_1n, I0n, I1n, filtI0n, size); }

// ---- SURVEY 8(f)1 building blocks of TV-L1 with occlusions ------------------------------------------------------------
void ref_median_filtering(double *in, int nx, int ny, int wsize) { me_median_filtering(in, nx, ny, wsize); }

void ref_occ_solver_v(const double *u1, const double *u2, double *v1, double *v2, const double *chi, const double *I1wx,
                      const double *I1wy, const double *I_1wx, const double *I_1wy, const double *rho1_c,
                      const double *rho3_c, double *Vfwd_1, double *Vfwd_2, double *Vbck_1, double *Vbck_2,
                      const double *grad1, const double *grad3, double alpha, double theta, double lambda, int nx, int ny)
{
    Solver_wrt_v(const_cast<double *>(u1), const_cast<double *>(u2), v1, v2, const_cast<double *>(chi), I1wx, I1wy, I_1wx,
                 I_1wy, rho1_c, rho3_c, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2, grad1, grad3, alpha, theta, lambda, nx, ny);
}

void ref_rof_box(double *u, const double *f, double *P1, double *P2, const double *g, double lambda, double omega, int nx,
                 int ny, int n_iter)
{ Scalar_ROF_BoxCellCentered(u, f, P1, P2, g, lambda, omega, nx, ny, n_iter); }

// One call of the reference's Solver_wrt_u (10 box-relaxation iterations per flow component).  Its dual planes are
// function-local statics, zeroed whenever nx differs from the previous call's: `fresh` != 0 first makes a dummy call
// with another width, so that the real call starts from p = 0; fresh == 0 continues with what the previous call left.
void ref_occ_solver_u(double *u1, double *u2, const double *v1, const double *v2, const double *chi, const double *g,
                      double theta, double beta, int nx, int ny, int fresh)
{
    static int last_nx = 0;                      // the width the function's statics are sized for (0 = never called)
    if (fresh) {
        int dn = 3;
        while (dn == nx || dn == last_nx) dn++;  // a width that forces a re-allocation now AND at the real call
        double z[25 * 6] = {0};
        for (int k = 0; k < 25; k++) z[5 * 25 + k] = 1.0;            // g = 1
        Solver_wrt_u(z, z + 25, z + 50, z + 75, z + 100, z + 125, theta, beta, dn, dn);
    }
    Solver_wrt_u(u1, u2, v1, v2, chi, g, theta, beta, nx, ny);
    last_nx = nx;
}

// One call of the reference's Solver_wrt_chi (MAX_ITERATIONS_CHI = 100 iterations).  Its dual variable lives in
// function-local statics that are re-allocated -- zero-filled, see operator new[] above -- whenever nx differs from the
// previous call's: `fresh` != 0 first makes a dummy call with another width, so that the real call starts from eta = 0;
// fresh == 0 continues with the eta the previous call left (same nx).
void ref_occ_solver_chi(const double *u1, const double *u2, double *chi, const double *I1wx, const double *I1wy,
                        const double *I_1wx, const double *I_1wy, const double *rho1_c, const double *rho3_c,
                        const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1, const double *Vbck_2,
                        const double *g, double lambda, double theta, double alpha, double beta, double tau_chi,
                        double tau_eta, int nx, int ny, int fresh)
{
    static int last_nx = 0;                      // the width the function's statics are sized for (0 = never called)
    if (fresh) {
        int dn = 3;
        while (dn == nx || dn == last_nx) dn++;  // a width that forces a re-allocation now AND at the real call
        double z[25 * 14] = {0};
        double *a = z;
        Solver_wrt_chi(a, a + 25, a + 50, a + 75, a + 100, a + 125, a + 150, a + 175, a + 200, a + 225, a + 250, a + 275,
                       a + 300, a + 325, lambda, theta, alpha, beta, tau_chi, tau_eta, dn, dn);
    }
    Solver_wrt_chi(u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2, g, lambda, theta,
                   alpha, beta, tau_chi, tau_eta, nx, ny);
    last_nx = nx;
}

} // extern "C"
