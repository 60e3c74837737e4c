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
#include "tvl1occflow.h"
#include "robust_expo_methods.h"
#include "robust_expo_smoothness.h"

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

// robust_expo_methods (robust_expo_methods.h:21-38), one channel; the multiscale entry.  The reference reports its sweep counts
// only on stdout (verbose), so they are not returned here.
int ref_robust_expo(const double *I1, const double *I2, double *u, double *v, int nx, int ny, int nz, int method_type,
                    double alpha, double gamma, double lambda, int nscales, double nu, double TOL, int inner_iter,
                    int outer_iter, int verbose)
{
    try {
        robust_expo_methods(I1, I2, u, v, nx, ny, nz, method_type, alpha, gamma, lambda, nscales, nu, TOL, inner_iter,
                            outer_iter, verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    return 0;
}
void ref_rexpo_exponential(const double *Ix, const double *Iy, int size_flow, int size, int nz, double alpha, double lambda,
                           int method_type, double *expo)
{ robust_expo_exponential_calculation(Ix, Iy, size_flow, size, nz, alpha, lambda, method_type, expo); }
void ref_gaussian_bc(double *I, int nx, int ny, double sigma, int bc) { gaussian(I, nx, ny, sigma, bc); }

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

// ---- the two stateful solvers ---------------------------------------------------------------------------------------
// Solver_wrt_u keeps its four dual planes, Solver_wrt_chi its dual variable, in function-local statics that are
// re-allocated (and thereby zeroed: Solver_wrt_u does it itself, Solver_wrt_chi through operator new[] above) only when nx
// differs from the previous call's.  occ_reset() forces that for the NEXT call of width nx, whatever ran before: a first
// dummy call on 2 x 2 (the smallest image: if the statics happen to be 2 wide already they hold at least 4 values), a
// second one on a width that is neither 2 nor nx.
static void occ_reset_u(int nx, double theta, double beta)
{
    double z[10 * 6] = {0};
    for (int k = 0; k < 10; k++) z[5 * 10 + k] = 1.0;                // g = 1
    const int w = (nx == 3) ? 5 : 3;
    Solver_wrt_u(z, z + 10, z + 20, z + 30, z + 40, z + 50, theta, beta, 2, 2);
    Solver_wrt_u(z, z + 10, z + 20, z + 30, z + 40, z + 50, theta, beta, w, 2);
}
static void occ_reset_chi(int nx)
{
    double z[10 * 14] = {0};
    double *a = z;
    const int w = (nx == 3) ? 5 : 3;
    for (int pass = 0; pass < 2; pass++)
        Solver_wrt_chi(a, a + 10, a + 20, a + 30, a + 40, a + 50, a + 60, a + 70, a + 80, a + 90, a + 100, a + 110, a + 120,
                       a + 130, 0.15, 0.3, 0.01, 0.15, 0.15, 0.15, pass ? w : 2, 2);
}

// One call of the reference's Solver_wrt_u (10 box-relaxation iterations per flow component).  fresh != 0: start from
// zero dual planes; fresh == 0: continue with what the previous call (same nx) left.
void ref_occ_solver_u(double *u1, double *u2, const double *v1, const double *v2, const double *chi, const double *g,
                      double theta, double beta, int nx, int ny, int fresh)
{
    if (fresh) occ_reset_u(nx, theta, beta);
    Solver_wrt_u(u1, u2, v1, v2, chi, g, theta, beta, nx, ny);
}

// One call of the reference's Solver_wrt_chi (MAX_ITERATIONS_CHI = 100 iterations); fresh as above (dual variable = 0)
void ref_occ_solver_chi(const double *u1, const double *u2, double *chi, const double *I1wx, const double *I1wy,
                        const double *I_1wx, const double *I_1wy, const double *rho1_c, const double *rho3_c,
                        const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1, const double *Vbck_2,
                        const double *g, double lambda, double theta, double alpha, double beta, double tau_chi,
                        double tau_eta, int nx, int ny, int fresh)
{
    if (fresh) occ_reset_chi(nx);
    Solver_wrt_chi(u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vfwd_1, Vfwd_2, Vbck_1, Vbck_2, g, lambda, theta,
                   alpha, beta, tau_chi, tau_eta, nx, ny);
}

// The whole TV-L1-with-occlusions solve (src/tvl1occflow.h): the statics of both solvers are reset first, so that the
// coarsest level starts from zero dual state whatever ran before -- every finer level resets them itself (its width
// differs from the previous level's).  Together with the zero-filling operator new[] this makes the reference program
// deterministic: "what it computes on a zero-filled heap".
void ref_tvl1occ_multiscale(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1, double *u2,
                            double *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta, int nscales,
                            double zfactor, int warps, double epsilon, int verbose)
{
    int nxc = nxx, nyc = nyy;
    for (int s = 1; s < nscales; s++) zoom_size(nxc, nyc, &nxc, &nyc, zfactor);
    occ_reset_u(nxc, theta, beta);
    occ_reset_chi(nxc);
    Dual_TVL1_optic_flow_multiscale(const_cast<double *>(I_1), const_cast<double *>(I0), const_cast<double *>(I1),
                                    const_cast<double *>(filtI0), u1, u2, chi, nxx, nyy, lambda, alpha, beta, theta, nscales,
                                    zfactor, warps, epsilon, verbose != 0);
}

} // extern "C"
