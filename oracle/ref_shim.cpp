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

#ifdef _OPENMP
#include <omp.h>
#endif

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

} // extern "C"
