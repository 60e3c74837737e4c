// ofx_api.cpp -- host-pointer entry points of the operator-level C ABI (include/ofx.h).
// Each call: upload double planes -> (convert to the storage type) -> kernel(s) -> convert back ->
// download.  These exist for drop-in / parity use; the solvers never round-trip through the host.
#include "ofx_ops.h"

#include <vector>

namespace {

template <typename T> struct HostOp {
    ofx_ctx *ctx;
    size_t n;
    std::vector<double *> stage;

    // device image of storage type T filled from a host double plane
    int in(const double *h, T **out, size_t count)
    {
        double *s;
        OFX_TRY(ofx_alloc(ctx, count, &s));
        OFX_TRY(ofx_alloc(ctx, count, out));
        OFX_HIP(ctx, hipMemcpyAsync(s, h, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        return op_convert_in<T>(ctx, s, *out, count);
    }
    int out_alloc(T **out, size_t count) { return ofx_alloc(ctx, count, out); }
    // device image -> host double plane
    int out(const T *d, double *h, size_t count)
    {
        double *s;
        OFX_TRY(ofx_alloc(ctx, count, &s));
        OFX_TRY(op_convert_out<T>(ctx, d, s, count));
        OFX_HIP(ctx, hipMemcpyAsync(h, s, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        return OFX_OK;
    }
    int sync()
    {
        OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return OFX_OK;
    }
};

int check_dims(ofx_ctx *ctx, int nx, int ny)
{
    if (nx < 2 || ny < 2 || (long long) nx * ny > 0x7fffffffLL) return ofx_fail(ctx, OFX_ERR_ARG, "bad image size %dx%d", nx, ny);
    return OFX_OK;
}

template <typename T> int divergence_t(ofx_ctx *ctx, const double *v1, const double *v2, double *div, int nx, int ny)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a, *b, *o;
    OFX_TRY(h.in(v1, &a, n));
    OFX_TRY(h.in(v2, &b, n));
    OFX_TRY(h.out_alloc(&o, n));
    OFX_TRY(op_divergence<T>(ctx, a, b, o, nx, ny));
    OFX_TRY(h.out(o, div, n));
    return h.sync();
}

template <typename T>
int gradient_t(ofx_ctx *ctx, const double *f, double *gx, double *gy, int nx, int ny, bool centered)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a, *ox, *oy;
    OFX_TRY(h.in(f, &a, n));
    OFX_TRY(h.out_alloc(&ox, n));
    OFX_TRY(h.out_alloc(&oy, n));
    if (centered) OFX_TRY(op_centered_gradient<T>(ctx, a, ox, oy, nx, ny));
    else OFX_TRY(op_forward_gradient<T>(ctx, a, ox, oy, nx, ny));
    OFX_TRY(h.out(ox, gx, n));
    OFX_TRY(h.out(oy, gy, n));
    return h.sync();
}

template <typename T> int second_t(ofx_ctx *ctx, const double *f, double *out, int nx, int ny, int which)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a, *o;
    OFX_TRY(h.in(f, &a, n));
    OFX_TRY(h.out_alloc(&o, n));
    OFX_TRY(op_second_derivative<T>(ctx, a, o, nx, ny, which));
    OFX_TRY(h.out(o, out, n));
    return h.sync();
}

template <typename T> int gaussian_t(ofx_ctx *ctx, double *I, int nx, int ny, double sigma)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a, *tmp;
    OFX_TRY(h.in(I, &a, n));
    OFX_TRY(h.out_alloc(&tmp, n));
    OFX_TRY(op_gaussian<T>(ctx, a, tmp, nx, ny, sigma));
    OFX_TRY(h.out(a, I, n));
    return h.sync();
}

template <typename T>
int warp_t(ofx_ctx *ctx, const double *in, const double *u, const double *v, double *out, int nx, int ny, int bo)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a, *du, *dv, *o;
    OFX_TRY(h.in(in, &a, n));
    OFX_TRY(h.in(u, &du, n));
    OFX_TRY(h.in(v, &dv, n));
    OFX_TRY(h.out_alloc(&o, n));
    OFX_TRY(op_bicubic_warp<T>(ctx, a, du, dv, o, nx, ny, bo));
    OFX_TRY(h.out(o, out, n));
    return h.sync();
}

template <typename T>
int at_t(ofx_ctx *ctx, const double *in, const double *uu, const double *vv, double *out, int cnt, int nx, int ny, int bo)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    T *a;
    double *du, *dv, *o;
    OFX_TRY(h.in(in, &a, n));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &du));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &dv));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &o));
    OFX_HIP(ctx, hipMemcpyAsync(du, uu, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(dv, vv, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_TRY(op_bicubic_at<T>(ctx, a, du, dv, o, cnt, nx, ny, bo));
    OFX_HIP(ctx, hipMemcpyAsync(out, o, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return h.sync();
}

template <typename T> int zoom_out_t(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, double factor)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny;
    int nxx, nyy;
    ofx_zoom_size(nx, ny, &nxx, &nyy, factor);
    if (nxx < 1 || nyy < 1) return ofx_fail(ctx, OFX_ERR_ARG, "zoom_out: empty output");
    T *a, *o, *t1, *t2;
    OFX_TRY(h.in(I, &a, n));
    OFX_TRY(h.out_alloc(&o, (size_t) nxx * nyy));
    OFX_TRY(h.out_alloc(&t1, n));
    OFX_TRY(h.out_alloc(&t2, n));
    OFX_TRY(op_zoom_out<T>(ctx, a, o, t1, t2, nx, ny, factor));
    OFX_TRY(h.out(o, Iout, (size_t) nxx * nyy));
    return h.sync();
}

template <typename T> int zoom_in_t(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, int nxx, int nyy)
{
    HostOp<T> h{ctx};
    T *a, *o;
    OFX_TRY(h.in(I, &a, (size_t) nx * ny));
    OFX_TRY(h.out_alloc(&o, (size_t) nxx * nyy));
    OFX_TRY(op_resample<T>(ctx, a, o, nx, ny, nxx, nyy, ((double) nxx / nx), ((double) nyy / ny)));
    OFX_TRY(h.out(o, Iout, (size_t) nxx * nyy));
    return h.sync();
}

template <typename T> int norm_t(ofx_ctx *ctx, const double *I1, const double *I2, double *o1, double *o2, int size)
{
    HostOp<T> h{ctx};
    T *a, *b, *x, *y;
    double *scr;
    OFX_TRY(h.in(I1, &a, (size_t) size));
    OFX_TRY(h.in(I2, &b, (size_t) size));
    OFX_TRY(h.out_alloc(&x, (size_t) size));
    OFX_TRY(h.out_alloc(&y, (size_t) size));
    OFX_TRY(ofx_alloc(ctx, (size_t) 2 * 1024 + 2, &scr));
    OFX_TRY(op_normalize2<T>(ctx, a, b, x, y, size, scr));
    OFX_TRY(h.out(x, o1, (size_t) size));
    OFX_TRY(h.out(y, o2, (size_t) size));
    return h.sync();
}

template <typename T>
int at_color_t(ofx_ctx *ctx, const double *in, const double *uu, const double *vv, double *out, int cnt, int nx, int ny,
               int nz, int ch, int bo)
{
    HostOp<T> h{ctx};
    const size_t n = (size_t) nx * ny * nz;
    T *a;
    double *du, *dv, *o;
    OFX_TRY(h.in(in, &a, n));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &du));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &dv));
    OFX_TRY(ofx_alloc(ctx, (size_t) cnt, &o));
    OFX_HIP(ctx, hipMemcpyAsync(du, uu, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_HIP(ctx, hipMemcpyAsync(dv, vv, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    OFX_TRY(op_bicubic_at_color<T>(ctx, a, du, dv, o, cnt, nx, ny, nz, ch, bo));
    OFX_HIP(ctx, hipMemcpyAsync(out, o, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return h.sync();
}

template <typename T>
int gradient3_t(ofx_ctx *ctx, const double *f, double *gx, double *gy, double *gz, int nx, int ny, int nz)
{
    HostOp<T> h{ctx};
    const size_t df = (size_t) nx * ny, n = df * nz;
    T *a, *ox, *oy, *oz;
    OFX_TRY(h.in(f, &a, n));
    OFX_TRY(h.out_alloc(&ox, n));
    OFX_TRY(h.out_alloc(&oy, n));
    OFX_TRY(h.out_alloc(&oz, n));
    for (int k = 0; k < nz; k++) OFX_TRY(op_centered_gradient<T>(ctx, a + k * df, ox + k * df, oy + k * df, nx, ny));
    OFX_TRY(op_gradient_dz<T>(ctx, a, oz, nx, ny, nz));
    OFX_TRY(h.out(ox, gx, n));
    OFX_TRY(h.out(oy, gy, n));
    OFX_TRY(h.out(oz, gz, n));
    return h.sync();
}

template <typename T> int norm1_t(ofx_ctx *ctx, const double *I, double *o, int size)
{
    HostOp<T> h{ctx};
    T *a, *x, *dummy;
    double *scr;
    OFX_TRY(h.in(I, &a, (size_t) size));
    OFX_TRY(h.out_alloc(&x, (size_t) size));
    OFX_TRY(h.out_alloc(&dummy, (size_t) size));
    OFX_TRY(ofx_alloc(ctx, op_pyramid_scratch_doubles(), &scr));
    OFX_TRY(op_normalize2<T>(ctx, a, a, x, dummy, size, scr));      // joint min / max of (I, I) = min / max of I
    OFX_TRY(h.out(x, o, (size_t) size));
    return h.sync();
}

template <typename T> int minmax_t(ofx_ctx *ctx, const double *x, int size, double *mn, double *mx)
{
    HostOp<T> h{ctx};
    T *a;
    double *scr, mm[2];
    OFX_TRY(h.in(x, &a, (size_t) size));
    OFX_TRY(ofx_alloc(ctx, op_pyramid_scratch_doubles(), &scr));
    OFX_TRY(op_minmax<T>(ctx, a, size, scr));
    OFX_HIP(ctx, hipMemcpyAsync(mm, scr + 2 * 1024, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFX_TRY(h.sync());
    *mn = mm[0];
    *mx = mm[1];
    return OFX_OK;
}

} // namespace

#define DISPATCH(ctx, fn, ...) ((ctx)->precision == OFX_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

extern "C" {

void ofx_zoom_size(int nx, int ny, int *nxx, int *nyy, double factor)      // src/zoom.cpp:22-34
{
    *nxx = (int) (nx * factor + 0.5);
    *nyy = (int) (ny * factor + 0.5);
}

int ofx_divergence(ofx_ctx *ctx, const double *v1, const double *v2, double *div, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!v1 || !v2 || !div) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, divergence_t, ctx, v1, v2, div, nx, ny);
}

int ofx_forward_gradient(ofx_ctx *ctx, const double *f, double *fx, double *fy, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!f || !fx || !fy) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, gradient_t, ctx, f, fx, fy, nx, ny, false);
}

int ofx_centered_gradient(ofx_ctx *ctx, const double *f, double *dx, double *dy, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!f || !dx || !dy) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, gradient_t, ctx, f, dx, dy, nx, ny, true);
}

int ofx_dxx(ofx_ctx *ctx, const double *I, double *o, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!I || !o) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, second_t, ctx, I, o, nx, ny, 0);
}

int ofx_dyy(ofx_ctx *ctx, const double *I, double *o, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!I || !o) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, second_t, ctx, I, o, nx, ny, 1);
}

int ofx_dxy(ofx_ctx *ctx, const double *I, double *o, int nx, int ny)
{
    OFX_ENTER(ctx);
    if (!I || !o) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, second_t, ctx, I, o, nx, ny, 2);
}

int ofx_gaussian(ofx_ctx *ctx, double *I, int nx, int ny, double sigma)
{
    OFX_ENTER(ctx);
    if (!I) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    if (!(sigma > 0)) return ofx_fail(ctx, OFX_ERR_ARG, "gaussian: sigma=%g", sigma);
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, gaussian_t, ctx, I, nx, ny, sigma);
}

int ofx_bicubic_at(ofx_ctx *ctx, const double *input, const double *uu, const double *vv, double *out, int n,
                   int nx, int ny, int border_out)
{
    OFX_ENTER(ctx);
    if (!input || !uu || !vv || !out || n < 1) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer / n < 1");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, at_t, ctx, input, uu, vv, out, n, nx, ny, border_out);
}

int ofx_bicubic_warp(ofx_ctx *ctx, const double *input, const double *u, const double *v, double *output, int nx,
                     int ny, int border_out)
{
    OFX_ENTER(ctx);
    if (!input || !u || !v || !output) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, warp_t, ctx, input, u, v, output, nx, ny, border_out);
}

int ofx_zoom_out(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, double factor)
{
    OFX_ENTER(ctx);
    if (!I || !Iout) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    if (!(factor > 0) || !(factor < 1)) return ofx_fail(ctx, OFX_ERR_ARG, "zoom_out: factor=%g", factor);
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, zoom_out_t, ctx, I, Iout, nx, ny, factor);
}

int ofx_zoom_in(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, int nxx, int nyy)
{
    OFX_ENTER(ctx);
    if (!I || !Iout) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    OFX_TRY(check_dims(ctx, nx, ny));
    if (nxx < 1 || nyy < 1) return ofx_fail(ctx, OFX_ERR_ARG, "zoom_in: bad output size");
    return DISPATCH(ctx, zoom_in_t, ctx, I, Iout, nx, ny, nxx, nyy);
}

int ofx_bicubic_at_color(ofx_ctx *ctx, const double *input, const double *uu, const double *vv, double *out, int n,
                         int nx, int ny, int nz, int k, int border_out)
{
    OFX_ENTER(ctx);
    if (!input || !uu || !vv || !out || n < 1) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer / n < 1");
    if (nz < 1 || k < 0 || k >= nz) return ofx_fail(ctx, OFX_ERR_ARG, "bicubic_at_color: channel %d of %d", k, nz);
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, at_color_t, ctx, input, uu, vv, out, n, nx, ny, nz, k, border_out);
}

int ofx_centered_gradient3(ofx_ctx *ctx, const double *f, double *dx, double *dy, double *dz, int nx, int ny, int nz)
{
    OFX_ENTER(ctx);
    if (!f || !dx || !dy || !dz) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer");
    if (nz < 1) return ofx_fail(ctx, OFX_ERR_ARG, "centered_gradient3: nz=%d", nz);
    OFX_TRY(check_dims(ctx, nx, ny));
    return DISPATCH(ctx, gradient3_t, ctx, f, dx, dy, dz, nx, ny, nz);
}

int ofx_zoom_out_color(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, int nz, double factor)
{
    // src/zoom.cpp:85-125 copies and smooths only nx*ny samples of the nz-channel image and then indexes that
    // scratch copy with interleaved-channel offsets: for nz > 1 it reads beyond the allocation (undefined
    // behaviour), for nz == 1 it IS zoom_out.  Only the defined case is provided.
    if (!ctx) return OFX_ERR_ARG;
    if (nz != 1) return ofx_fail(ctx, OFX_ERR_ARG, "zoom_out_color: nz=%d (the reference is only defined for nz = 1)", nz);
    return ofx_zoom_out(ctx, I, Iout, nx, ny, factor);
}

int ofx_image_normalization_1(ofx_ctx *ctx, const double *I, double *In, int size)
{
    OFX_ENTER(ctx);
    if (!I || !In || size < 1) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer / size < 1");
    return DISPATCH(ctx, norm1_t, ctx, I, In, size);
}

int ofx_getminmax(ofx_ctx *ctx, const double *x, int n, double *min, double *max)
{
    OFX_ENTER(ctx);
    if (!x || !min || !max || n < 1) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer / n < 1");
    return DISPATCH(ctx, minmax_t, ctx, x, n, min, max);
}

int ofx_image_normalization_2(ofx_ctx *ctx, const double *I1, const double *I2, double *I1n, double *I2n, int size)
{
    OFX_ENTER(ctx);
    if (!I1 || !I2 || !I1n || !I2n || size < 1) return ofx_fail(ctx, OFX_ERR_ARG, "NULL pointer / size < 1");
    return DISPATCH(ctx, norm_t, ctx, I1, I2, I1n, I2n, size);
}

} // extern "C"
