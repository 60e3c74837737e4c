// ofx_ops.h -- launchers of the operator / pyramid kernels (device pointers, enqueue on ctx->stream,
// never synchronise).  T is the storage type (double | float); arithmetic is double.
#pragma once

#include "ofx_internal.h"

#include <vector>

#define OFX_GAUSS_MAX_TAPS 64

struct GaussTaps {
    int    size;                       // kernel radius + 1  == (int)(5 sigma) + 1
    int    dirichlet;                  // 1: samples outside the image are 0 (BOUNDARY_CONDITION_DIRICHLET); 0: reflecting (default)
    double B[OFX_GAUSS_MAX_TAPS];
};

// host-side tap computation (src/operators.cpp:515-539); returns OFX_ERR_ARG if > MAX taps
int ofx_gauss_taps(double sigma, GaussTaps *t);

template <typename T> int op_convert_in(ofx_ctx *ctx, const double *src, T *dst, size_t n);
template <typename T> int op_convert_out(ofx_ctx *ctx, const T *src, double *dst, size_t n);
template <typename T> int op_interleave2(ofx_ctx *ctx, const double *a, const double *b,
                                         typename Pix<T>::v2 *dst, size_t n);
template <typename T> int op_deinterleave2(ofx_ctx *ctx, const typename Pix<T>::v2 *src, double *a,
                                           double *b, size_t n);
template <typename T> int op_to_flo(ofx_ctx *ctx, const typename Pix<T>::v2 *src, float2 *dst, size_t n);
template <typename T> int op_fill2(ofx_ctx *ctx, typename Pix<T>::v2 *dst, size_t n);   // zero

// image_normalization_2 (src/utils.cpp:283-326); scratch: 4 * 1024 T-sized... see .hip (own alloc)
template <typename T> int op_normalize2(ofx_ctx *ctx, const T *I1, const T *I2, T *o1, T *o2, int size,
                                        double *d_scratch /* >= 2*2048+2 doubles */);

// gaussian (src/operators.cpp:506-624): I updated in place, tmp is an nx*ny scratch image
template <typename T> int op_gaussian(ofx_ctx *ctx, T *I, T *tmp, int nx, int ny, double sigma, int dirichlet = 0);

// generic bicubic resampling out(i1,j1) = bicubic(in, j1/fx, i1/fy, border_out=false)
// (zoom_out: fx=fy=factor, src/zoom.cpp:67-75; zoom_in: per-axis factors, src/zoom.cpp:142-154)
template <typename T> int op_resample(ofx_ctx *ctx, const T *in, T *out, int nx, int ny, int nxx, int nyy,
                                      double fx, double fy);
// zoom_in of an interleaved flow field + `*= scale` (src/tvl1flow.cpp:302-309)
template <typename T> int op_zoom_in_flow(ofx_ctx *ctx, const typename Pix<T>::v2 *U, typename Pix<T>::v2 *Uout,
                                          int nx, int ny, int nxx, int nyy, double scale);
// zoom_out (src/zoom.cpp:41-78): tmpA, tmpB are nx*ny scratch images
template <typename T> int op_zoom_out(ofx_ctx *ctx, const T *I, T *Iout, T *tmpA, T *tmpB, int nx, int ny,
                                      double factor);

// planar operators for the operator-level API
template <typename T> int op_divergence(ofx_ctx *ctx, const T *v1, const T *v2, T *div, int nx, int ny);
template <typename T> int op_forward_gradient(ofx_ctx *ctx, const T *f, T *fx, T *fy, int nx, int ny);
template <typename T> int op_centered_gradient(ofx_ctx *ctx, const T *f, T *dx, T *dy, int nx, int ny);
template <typename T> int op_second_derivative(ofx_ctx *ctx, const T *f, T *out, int nx, int ny, int which);
template <typename T> int op_bicubic_warp(ofx_ctx *ctx, const T *in, const T *u, const T *v, T *out, int nx,
                                          int ny, int border_out);
template <typename T> int op_bicubic_at(ofx_ctx *ctx, const T *in, const double *uu, const double *vv,
                                        double *out, int n, int nx, int ny, int border_out);

// colour / sequence variants of the operator surface (bicubic_interpolation.h:30, operators.h:107, utils.h:119)
template <typename T> int op_bicubic_at_color(ofx_ctx *ctx, const T *in, const double *uu, const double *vv, double *out,
                                              int n, int nx, int ny, int nz, int ch, int border_out);
template <typename T> int op_gradient_dz(ofx_ctx *ctx, const T *in, T *dz, int nx, int ny, int nz);
// scr >= op_pyramid_scratch_doubles(); result: scr[2 * 1024] = min, scr[2 * 1024 + 1] = max
template <typename T> int op_minmax(ofx_ctx *ctx, const T *x, int size, double *scr);

// pa = (f, centred dx) pairs, pb = centred dy: one pair gather + one scalar gather per bicubic tap serve the
// three warps of (f, fx, fy) (see bicubic_sample3 in ofx_device.h for why not one padded 4-vector)
template <typename T> int op_grad_pack(ofx_ctx *ctx, const T *f, typename Pix<T>::v2 *pa, T *pb, int nx, int ny, int G = 1);

// Shared pyramid prologue (src/tvl1flow.cpp:236-280 == horn_schunck_pyramidal.cpp:279-323 ==
// brox_optic_flow_spatial.cpp:467-509): joint normalisation, presmoothing, zoom_out chain.
template <typename T> struct ImgLevel {
    int nx, ny;
    T *A, *B;
};
template <typename T>
int op_build_pyramid(ofx_ctx *ctx, const T *dA, const T *dB, int nx, int ny, int nscales, double zfactor,
                     double presmooth_sigma, std::vector<ImgLevel<T>> &lv);

// the same in pieces, for callers that lay several pyramids out in their own arrays (TV-L1 lockstep groups)
int op_pyramid_sizes(ofx_ctx *ctx, int nxx, int nyy, int nscales, double zfactor, std::vector<int> &nxs, std::vector<int> &nys);
size_t op_pyramid_scratch_doubles();
template <typename T>
int op_build_pyramid_into(ofx_ctx *ctx, const T *dA, const T *dB, int nscales, double zfactor, double presmooth_sigma,
                          const std::vector<ImgLevel<T>> &lv, T *tmpA, T *tmpB, double *scr);

// device pointers of the caller's G image pairs, passed to kernels by value
struct OfxGroupPtrs {
    const void *a[OFX_MAX_GROUP];
    const void *b[OFX_MAX_GROUP];
};
// pyramids of the G pairs of a lockstep group, one launch per step for all 2 G images (see ofx_ops.hip)
template <typename T>
int op_build_pyramid_group(ofx_ctx *ctx, int G, const void *const *dA, const void *const *dB, int nscales, double zfactor,
                           double presmooth_sigma, const int *nxs, const int *nys, T *const *lvA, T *const *lvB, T *tmpA,
                           T *tmpB, double *scr);

#define OFX_LAUNCH_CHECK(ctx)                                                                    \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess)                                                                   \
            return ofx_fail((ctx), OFX_ERR_HIP, "kernel launch failed: %s (%s:%d)",              \
                            hipGetErrorString(e__), __FILE__, __LINE__);                         \
    } while (0)
