// ofx_reference_shim.hpp -- the reference's OWN prototypes, defined on top of libofx.so (include/ofx.h).
//
// A program written against the reference headers (src/tvl1flow.h, tvl1occflow.h, tvl1occflow_solvers.h, horn_schunck.h,
// brox_optic_flow.h, operators.h, bicubic_interpolation.h, zoom.h, utils.h -- cited per function below) keeps compiling unchanged: include this file in
// ONE translation unit instead of compiling tvl1flow.cpp, operators.cpp, bicubic_interpolation.cpp, zoom.cpp,
// utils.cpp, horn_schunck_pyramidal.cpp, brox_optic_flow_spatial.cpp, brox_spatial_mask.cpp,
// brox_optic_flow_temporal.cpp, brox_temporal_mask.cpp, and link with -lofx.  `ofpix_t` must be double (src/of.h:4-10,
// the reference default).  Failures surface the way the reference's do: std::runtime_error("GaussianSmooth: sigma
// too large") (src/operators.cpp:520-522), std::bad_alloc; everything else (no GPU, HIP error, an argument only the
// multi-channel CPU code accepts) is a std::runtime_error with the library's message.  There is no CPU fallback.
// tests/shim/shim_driver.cpp is compiled against the reference's real headers with this file and run on the GPU by
// tests/test_gpu_shim.py.
#ifndef OFX_REFERENCE_SHIM_HPP
#define OFX_REFERENCE_SHIM_HPP

#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>

#include "ofx.h"

#include "bicubic_interpolation.h"
#include "brox_optic_flow.h"
#include "horn_schunck.h"
#include "operators.h"
#include "robust_expo_methods.h"
#include "tvl1flow.h"
#include "tvl1occflow.h"
#include "tvl1occflow_solvers.h"
#include "utils.h"
#include "zoom.h"

namespace ofx_shim {

// one context per process: device OFX_DEVICE (default 0), strict double storage
inline ofx_ctx *ctx()
{
    static ofx_ctx *c = nullptr;
    if (!c) {
        const char *d = std::getenv("OFX_DEVICE");
        if (ofx_ctx_create(&c, d ? std::atoi(d) : 0, OFX_F64) != OFX_OK) throw std::runtime_error("libofx: no usable gfx950 device");
    }
    return c;
}
inline void check(int s)
{
    if (s == OFX_OK) return;
    if (s == OFX_ERR_NOMEM) throw std::bad_alloc();
    if (s == OFX_ERR_SIGMA) throw std::runtime_error("GaussianSmooth: sigma too large");
    throw std::runtime_error(std::string("libofx: ") + ofx_strerror(s) + " (" + ofx_last_error(ctx()) + ")");
}
inline void single_channel(int nz, const char *fn)
{
    if (nz != 1) throw std::runtime_error(std::string("libofx: ") + fn + " is provided for nz = 1 only");
}

} // namespace ofx_shim

// ---- src/tvl1flow.h:36-70 ---------------------------------------------------------------------------------------
void Dual_TVL1_optic_flow(ofpix_t *I0, ofpix_t *I1, ofpix_t *u1, ofpix_t *u2, const int nx, const int ny, const double tau,
                          const double lambda, const double theta, const int warps, const double epsilon, const bool verbose)
{
    ofx_shim::check(ofx_tvl1_single_scale(ofx_shim::ctx(), I0, I1, u1, u2, nx, ny, tau, lambda, theta, warps, epsilon, verbose));
}
void Dual_TVL1_optic_flow_multiscale(ofpix_t *I0, ofpix_t *I1, ofpix_t *u1, ofpix_t *u2, const int nxx, const int nyy,
                                     const double tau, const double lambda, const double theta, const int nscales,
                                     const double zfactor, const int warps, const double epsilon, const bool verbose)
{
    ofx_shim::check(ofx_tvl1_multiscale(ofx_shim::ctx(), I0, I1, u1, u2, nxx, nyy, tau, lambda, theta, nscales, zfactor, warps,
                                        epsilon, verbose));
}

// ---- src/horn_schunck.h:15-48 -----------------------------------------------------------------------------------
void horn_schunck_optical_flow(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *u, ofpix_t *v, const int nx, const int ny,
                               const double alpha, const int warps, const double TOL, const int maxiter, const bool verbose)
{
    ofx_shim::check(ofx_hs_single_scale(ofx_shim::ctx(), I1, I2, u, v, nx, ny, alpha, warps, TOL, maxiter, verbose));
}
void horn_schunck_pyramidal(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *u, ofpix_t *v, const int nx, const int ny,
                            const double alpha, const int nscales, const double zfactor, const int warps, const double TOL,
                            const int maxiter, const bool verbose)
{
    ofx_shim::check(ofx_hs_pyramidal(ofx_shim::ctx(), I1, I2, u, v, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter, verbose));
}

void hs(ofpix_t *u, ofpix_t *v, ofpix_t *a, ofpix_t *b, int w, int h, int n, double alpha)      // src/horn_schunck.h:7-8
{
    ofx_shim::check(ofx_hs_classic(ofx_shim::ctx(), a, b, u, v, w, h, n, alpha));
}

// ---- src/brox_optic_flow.h:19-55 --------------------------------------------------------------------------------
void brox_optic_flow_spatial(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *u, ofpix_t *v, const int nxx, const int nyy,
                             const double alpha, const double gamma, const int nscales, const double nu, const double TOL,
                             const int inner_iter, const int outer_iter, const bool verbose)
{
    ofx_shim::check(ofx_brox_spatial(ofx_shim::ctx(), I1, I2, u, v, nxx, nyy, alpha, gamma, nscales, nu, TOL, inner_iter, outer_iter,
                                     verbose));
}
// ---- src/robust_expo_methods.h:21-38 (one channel; nzz != 1 -> std::runtime_error with the library's message) -----------
void robust_expo_methods(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *u, ofpix_t *v, const int nxx, const int nyy, const int nzz,
                         const int method_type, const double alpha, const double gamma, const double lambda, const int nscales,
                         const double nu, const double TOL, const int inner_iter, const int outer_iter, const bool verbose)
{
    ofx_shim::check(ofx_robust_expo(ofx_shim::ctx(), I1, I2, u, v, nxx, nyy, nzz, method_type, alpha, gamma, lambda, nscales, nu,
                                    TOL, inner_iter, outer_iter, verbose));
}
void brox_optic_flow_temporal(const ofpix_t *I, ofpix_t *u, ofpix_t *v, const int nxx, const int nyy, const int frames,
                              const double alpha, const double gamma, const int nscales, const double nu, const double TOL,
                              const int inner_iter, const int outer_iter, const bool verbose)
{
    if (frames <= 2) {                      // src/brox_optic_flow_temporal.cpp:537-541: message, no exception, u / v untouched
        fprintf(stderr, "The method needs more than two frames\n");
        return;
    }
    ofx_shim::check(ofx_brox_temporal(ofx_shim::ctx(), I, u, v, nxx, nyy, frames, alpha, gamma, nscales, nu, TOL, inner_iter,
                                      outer_iter, verbose));
}

// ---- src/operators.h:29-134 -------------------------------------------------------------------------------------
void divergence(const ofpix_t *v1, const ofpix_t *v2, ofpix_t *div, const int nx, const int ny)
{
    ofx_shim::check(ofx_divergence(ofx_shim::ctx(), v1, v2, div, nx, ny));
}
void forward_gradient(const ofpix_t *f, ofpix_t *fx, ofpix_t *fy, const int nx, const int ny)
{
    ofx_shim::check(ofx_forward_gradient(ofx_shim::ctx(), f, fx, fy, nx, ny));
}
void Dxx(const ofpix_t *I, ofpix_t *Ixx, const int nx, const int ny, const int nz)
{
    ofx_shim::single_channel(nz, "Dxx");
    ofx_shim::check(ofx_dxx(ofx_shim::ctx(), I, Ixx, nx, ny));
}
void Dyy(const ofpix_t *I, ofpix_t *Iyy, const int nx, const int ny, const int nz)
{
    ofx_shim::single_channel(nz, "Dyy");
    ofx_shim::check(ofx_dyy(ofx_shim::ctx(), I, Iyy, nx, ny));
}
void Dxy(const ofpix_t *I, ofpix_t *Ixy, const int nx, const int ny, const int nz)
{
    ofx_shim::single_channel(nz, "Dxy");
    ofx_shim::check(ofx_dxy(ofx_shim::ctx(), I, Ixy, nx, ny));
}
void centered_gradient(const ofpix_t *input, ofpix_t *dx, ofpix_t *dy, const int nx, const int ny, const int nz)
{
    ofx_shim::single_channel(nz, "centered_gradient");
    ofx_shim::check(ofx_centered_gradient(ofx_shim::ctx(), input, dx, dy, nx, ny));
}
void centered_gradient3(const ofpix_t *input, ofpix_t *dx, ofpix_t *dy, ofpix_t *dz, const int nx, const int ny, const int nz)
{
    ofx_shim::check(ofx_centered_gradient3(ofx_shim::ctx(), input, dx, dy, dz, nx, ny, nz));
}
void gaussian(ofpix_t *I, const int xdim, const int ydim, const double sigma, const int bc, const int precision)
{
    if (bc != DEFAULT_BOUNDARY_CONDITION || precision != DEFAULT_GAUSSIAN_WINDOW_SIZE)
        throw std::runtime_error("libofx: gaussian is provided with the default boundary condition and window only");
    ofx_shim::check(ofx_gaussian(ofx_shim::ctx(), I, xdim, ydim, sigma));
}

// ---- src/bicubic_interpolation.h:16-52 --------------------------------------------------------------------------
double bicubic_interpolation_at(const ofpix_t *input, const double uu, const double vv, const int nx, const int ny,
                                const bool border_out)
{
    double out = 0.0;
    ofx_shim::check(ofx_bicubic_at(ofx_shim::ctx(), input, &uu, &vv, &out, 1, nx, ny, border_out));
    return out;
}
double bicubic_interpolation_at_color(const ofpix_t *input, const double uu, const double vv, const int nx, const int ny,
                                      const int nz, const int k, const bool border_out)
{
    double out = 0.0;
    ofx_shim::check(ofx_bicubic_at_color(ofx_shim::ctx(), input, &uu, &vv, &out, 1, nx, ny, nz, k, border_out));
    return out;
}
void bicubic_interpolation_warp(const ofpix_t *input, const ofpix_t *u, const ofpix_t *v, ofpix_t *output, const int nx,
                                const int ny, bool border_out)
{
    ofx_shim::check(ofx_bicubic_warp(ofx_shim::ctx(), input, u, v, output, nx, ny, border_out));
}

// ---- src/zoom.h:20-63 -------------------------------------------------------------------------------------------
void zoom_size(int nx, int ny, int *nxx, int *nyy, double factor) { ofx_zoom_size(nx, ny, nxx, nyy, factor); }
void zoom_out(const ofpix_t *I, ofpix_t *Iout, const int nx, const int ny, const double factor)
{
    ofx_shim::check(ofx_zoom_out(ofx_shim::ctx(), I, Iout, nx, ny, factor));
}
void zoom_out_color(const ofpix_t *I, ofpix_t *Iout, const int nx, const int ny, const int nz, const double factor)
{
    ofx_shim::check(ofx_zoom_out_color(ofx_shim::ctx(), I, Iout, nx, ny, nz, factor));
}
void zoom_in(const ofpix_t *I, ofpix_t *Iout, int nx, int ny, int nxx, int nyy)
{
    ofx_shim::check(ofx_zoom_in(ofx_shim::ctx(), I, Iout, nx, ny, nxx, nyy));
}

// ---- src/utils.h:17-32,119 --------------------------------------------------------------------------------------
void image_normalization_1(const ofpix_t *I, ofpix_t *In, int size)
{
    ofx_shim::check(ofx_image_normalization_1(ofx_shim::ctx(), I, In, size));
}
void image_normalization_2(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *I1n, ofpix_t *I2n, int size)
{
    ofx_shim::check(ofx_image_normalization_2(ofx_shim::ctx(), I1, I2, I1n, I2n, size));
}
void getminmax(ofpix_t *min, ofpix_t *max, const ofpix_t *x, int n)
{
    ofx_shim::check(ofx_getminmax(ofx_shim::ctx(), x, n, min, max));
}

// ---- src/utils.h:10,39-87, src/bicubic_interpolation.h:59-67 (colour / three-frame operators) --------------------
void me_median_filtering(ofpix_t *in, int nx, int ny, int wsize)
{
    ofx_shim::check(ofx_me_median_filtering(ofx_shim::ctx(), in, nx, ny, wsize));
}
void image_normalization_2_color(const ofpix_t *I1, const ofpix_t *I2, ofpix_t *I1n, ofpix_t *I2n, int size, int nz)
{
    ofx_shim::check(ofx_image_normalization_2_color(ofx_shim::ctx(), I1, I2, I1n, I2n, size, nz));
}
void image_normalization_3(ofpix_t *I0, ofpix_t *I1, ofpix_t *I2, int size)
{
    ofx_shim::check(ofx_image_normalization_3(ofx_shim::ctx(), I0, I1, I2, size));
}
void image_normalization_4(const ofpix_t *I_1, const ofpix_t *I0, const ofpix_t *I1, const ofpix_t *filtI0, ofpix_t *I_1n,
                           ofpix_t *I0n, ofpix_t *I1n, ofpix_t *filtI0n, int size)
{
    ofx_shim::check(ofx_image_normalization_4(ofx_shim::ctx(), I_1, I0, I1, filtI0, I_1n, I0n, I1n, filtI0n, size));
}
void bicubic_interpolation_warp_color(const ofpix_t *input, const ofpix_t *u, const ofpix_t *v, ofpix_t *output, const int nx,
                                      const int ny, const int nz, bool border_out)
{
    ofx_shim::check(ofx_bicubic_warp_color(ofx_shim::ctx(), input, u, v, output, nx, ny, nz, border_out ? 1 : 0));
}

// ---- src/tvl1occflow.h (the multiscale overload), src/tvl1occflow_solvers.h:57-66 ----------------------------------
// The reference's Solver_wrt_u / Solver_wrt_chi keep hidden static state between calls (and read it uninitialised): their
// counterparts take that state as arguments (ofx_solver_wrt_u / ofx_solver_wrt_chi) and are not given the old signatures.
// The single-scale Dual_TVL1_optic_flow overload of tvl1occflow.h depends on the same hidden state and is not provided either.
void Dual_TVL1_optic_flow_multiscale(ofpix_t *I_1, ofpix_t *I0, ofpix_t *I1, ofpix_t *filtI0, ofpix_t *u1, ofpix_t *u2,
                                     ofpix_t *chi, const int nxx, const int nyy, const double lambda, const double alpha,
                                     const double beta, const double theta, const int nscales, const double zfactor,
                                     const int warps, const double epsilon, const bool verbose)
{
    ofx_shim::check(ofx_tvl1occ_multiscale(ofx_shim::ctx(), I_1, I0, I1, filtI0, u1, u2, chi, nxx, nyy, lambda, alpha, beta, theta,
                                           nscales, zfactor, warps, epsilon, verbose ? 1 : 0));
}
void Solver_wrt_v(ofpix_t *u1, ofpix_t *u2, ofpix_t *v1, ofpix_t *v2, ofpix_t *chi, const ofpix_t *I1wx, const ofpix_t *I1wy,
                  const ofpix_t *I_1wx, const ofpix_t *I_1wy, const ofpix_t *rho1_c, const ofpix_t *rho3_c, ofpix_t *Vfwd_1,
                  ofpix_t *Vfwd_2, ofpix_t *Vbck_1, ofpix_t *Vbck_2, const ofpix_t *grad1, const ofpix_t *grad3, const double alpha,
                  const double theta, const double lambda, const int nx, const int ny)
{
    ofx_shim::check(ofx_solver_wrt_v(ofx_shim::ctx(), u1, u2, v1, v2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, Vfwd_1, Vfwd_2,
                                     Vbck_1, Vbck_2, grad1, grad3, alpha, theta, lambda, nx, ny));
}

#endif
