/* oracle/ofx_oracle.c -- TEST INFRASTRUCTURE ONLY (see ofx_oracle.h).
 *
 * Plain-C restatement of the reference hot path, double storage + double arithmetic, same
 * association order as the reference expressions so that it is bit-identical to the compiled
 * reference with one OpenMP thread (checked in tests/test_oracle_vs_ref.py).  Border handling is
 * written as clamped / predicated index arithmetic wherever that is provably the same IEEE
 * expression as the reference's separate border loops; where the reference associates border
 * terms differently (divergence first/last column) the exact order is kept.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/).
 */
#include "ofx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TVL1_MAX_ITERATIONS   300      /* src/tvl1flow.cpp:22 */
#define TVL1_PRESMOOTH_SIGMA  0.8      /* src/tvl1flow.cpp:23 */
#define TVL1_GRAD_IS_ZERO     1E-10    /* src/tvl1flow.cpp:24 */
#define ZOOM_SIGMA_ZERO       0.6      /* src/zoom.cpp:15 */
#define GAUSS_WINDOW          5        /* src/operators.h:120 */
#define HS_SOR_W              1.9      /* src/horn_schunck_pyramidal.cpp:21 */
#define HS_PRESMOOTH_SIGMA    0.8      /* src/horn_schunck_pyramidal.cpp:22 */
#define BROX_EPSILON          0.001    /* src/brox_optic_flow_spatial.cpp:23 */
#define BROX_MAXITER          300      /* src/brox_optic_flow_spatial.cpp:24 */
#define BROX_SOR_W            1.9      /* src/brox_optic_flow_spatial.cpp:25 */
#define BROX_SIGMA            0.8      /* src/brox_optic_flow_spatial.cpp:26 */

static double *dalloc(size_t n)
{
    double *p = (double *) malloc((n ? n : 1) * sizeof(double));
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void) n;
#endif
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* src/operators.cpp:35-78  backward-difference divergence.
 * interior: (v1[p]-v1[p-1]) + (v2[p]-v2[p-nx]).  Rows 0 / ny-1 drop one v2 term and keep the
 * interior association; columns 0 / nx-1 are written by the reference as ((a + b) - c)
 * (operators.cpp:70-71), which is NOT the interior association, so they get their own branch. */
void orc_divergence(const double *v1, const double *v2, double *div, int nx, int ny)
{
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < ny; i++) {
        const int top = (i == 0), bot = (i == ny - 1);
        for (int j = 0; j < nx; j++) {
            const int p = i * nx + j;
            const int lef = (j == 0), rig = (j == nx - 1);
            double d;
            if (!lef && !rig) {
                const double dxv = v1[p] - v1[p - 1];
                if (top)      d = dxv + v2[p];                 /* :61 */
                else if (bot) d = dxv - v2[p - nx];            /* :62 */
                else          d = dxv + (v2[p] - v2[p - nx]);  /* :50-53 */
            } else if (lef) {
                if (top)      d = v1[p] + v2[p];               /* :74 */
                else if (bot) d = v1[p] - v2[p - nx];          /* :76 */
                else          d = v1[p] + v2[p] - v2[p - nx];  /* :70 */
            } else {
                if (top)      d = -v1[p - 1] + v2[p];              /* :75 */
                else if (bot) d = -v1[p - 1] - v2[p - nx];         /* :77 */
                else          d = -v1[p - 1] + v2[p] - v2[p - nx]; /* :71 */
            }
            div[p] = d;
        }
    }
}

/* src/operators.cpp:86-125  forward-difference gradient, zero across the right / bottom edge */
void orc_forward_gradient(const double *f, double *fx, double *fy, int nx, int ny)
{
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < ny; i++) {
        for (int j = 0; j < nx; j++) {
            const int p = i * nx + j;
            fx[p] = (j < nx - 1) ? f[p + 1] - f[p] : 0.0;
            fy[p] = (i < ny - 1) ? f[p + nx] - f[p] : 0.0;
        }
    }
}

/* src/operators.cpp:335-406 (nz = 1)  centred differences; at the border the missing neighbour
 * is the pixel itself and the factor stays 1/2 (:363-404) == clamped indices. */
void orc_centered_gradient(const double *f, double *dx, double *dy, int nx, int ny)
{
    #pragma omp parallel for
    for (int i = 0; i < ny; i++) {
        const int iu = (i > 0) ? i - 1 : 0, id = (i < ny - 1) ? i + 1 : ny - 1;
        for (int j = 0; j < nx; j++) {
            const int jl = (j > 0) ? j - 1 : 0, jr = (j < nx - 1) ? j + 1 : nx - 1;
            dx[i * nx + j] = 0.5 * (f[i * nx + jr] - f[i * nx + jl]);
            dy[i * nx + j] = 0.5 * (f[id * nx + j] - f[iu * nx + j]);
        }
    }
}

/* src/operators.cpp:132-256 mask3x3 (nz = 1).  Taps that fall outside are folded onto the edge
 * sample and their weights are summed BEFORE the multiply (e.g. first row: in[j-1]*(m0+m3), :168);
 * the products are accumulated in row-major order of the distinct clamped taps. */
static void mask3x3(const double *in, double *out, int nx, int ny, const double *m)
{
    #pragma omp parallel for
    for (int i = 0; i < ny; i++) {
        for (int j = 0; j < nx; j++) {
            int rr[3], cc[3];
            for (int l = 0; l < 3; l++) {
                int r = i + l - 1, c = j + l - 1;
                rr[l] = r < 0 ? 0 : (r > ny - 1 ? ny - 1 : r);
                cc[l] = c < 0 ? 0 : (c > nx - 1 ? nx - 1 : c);
            }
            double sum = 0;
            for (int l = 0; l < 3; l++) {
                if (l > 0 && rr[l] == rr[l - 1]) continue;           /* merged into an earlier row */
                for (int q = 0; q < 3; q++) {
                    if (q > 0 && cc[q] == cc[q - 1]) continue;       /* merged into an earlier col */
                    double w = 0;
                    int first = 1;
                    for (int l2 = 0; l2 < 3; l2++) {
                        if (rr[l2] != rr[l]) continue;
                        for (int q2 = 0; q2 < 3; q2++) {
                            if (cc[q2] != cc[q]) continue;
                            w = first ? m[l2 * 3 + q2] : w + m[l2 * 3 + q2];
                            first = 0;
                        }
                    }
                    sum += in[rr[l] * nx + cc[q]] * w;
                }
            }
            out[i * nx + j] = sum;
        }
    }
}

/* src/operators.cpp:263-328 */
void orc_dxx(const double *f, double *out, int nx, int ny)
{
    const double m[9] = { 0., 0., 0., 1., -2., 1., 0., 0., 0. };
    mask3x3(f, out, nx, ny, m);
}

void orc_dyy(const double *f, double *out, int nx, int ny)
{
    const double m[9] = { 0., 1., 0., 0., -2., 0., 0., 1., 0. };
    mask3x3(f, out, nx, ny, m);
}

void orc_dxy(const double *f, double *out, int nx, int ny)
{
    const double m[9] = { 1. / 4., 0., -1. / 4., 0., 0., 0., -1. / 4., 0., 1. / 4. };
    mask3x3(f, out, nx, ny, m);
}

/* src/operators.cpp:506-624, default arguments (reflecting boundary, window 5).
 * Kernel radius size = (int)(5 sigma) + 1 (:516); taps B[i] = 1/(sigma sqrt(2*3.1415926)) *
 * exp(-i*i/(2 sigma^2)) normalised by 2*sum(B) - B[0] (:524-539).  Row pass over the whole image,
 * then column pass on the row-smoothed image, both in place (:544-619).  Reflection (:557-562):
 * sample t < 0 reads I[-t] (edge not repeated); sample t >= n reads I[2n-1-t] (edge repeated).
 * The reference is serial here (no OpenMP) and so is this. */
static int gauss_taps(double sigma, double **Bout)
{
    const double den = 2 * sigma * sigma;
    const int size = (int) (GAUSS_WINDOW * sigma) + 1;
    double *B = dalloc((size_t) size);
    for (int i = 0; i < size; i++)
        B[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);
    double norm = 0;
    for (int i = 0; i < size; i++) norm += B[i];
    norm *= 2;
    norm -= B[0];
    for (int i = 0; i < size; i++) B[i] /= norm;
    *Bout = B;
    return size;
}

static inline int gauss_reflect(int t, int n)
{
    return t < 0 ? -t : (t >= n ? 2 * n - 1 - t : t);
}

int orc_gaussian(double *I, int nx, int ny, double sigma)
{
    double *B;
    const int size = gauss_taps(sigma, &B);
    if (size > nx) { free(B); return 1; }        /* :520-522 throws */
    /* The reference reads out of bounds when size >= ny or size == nx; mirror its column pass only
     * when that is safe and report the same error otherwise. */
    if (size >= ny || size >= nx) { free(B); return 1; }

    double *line = dalloc((size_t) (nx > ny ? nx : ny));
    for (int k = 0; k < ny; k++) {
        double *row = I + (size_t) k * nx;
        for (int x = 0; x < nx; x++) {
            double sum = B[0] * row[x];
            for (int j = 1; j < size; j++)
                sum += B[j] * (row[gauss_reflect(x - j, nx)] + row[gauss_reflect(x + j, nx)]);
            line[x] = sum;
        }
        memcpy(row, line, (size_t) nx * sizeof(double));
    }
    for (int k = 0; k < nx; k++) {
        for (int y = 0; y < ny; y++) {
            double sum = B[0] * I[(size_t) y * nx + k];
            for (int j = 1; j < size; j++)
                sum += B[j] * (I[(size_t) gauss_reflect(y - j, ny) * nx + k] +
                               I[(size_t) gauss_reflect(y + j, ny) * nx + k]);
            line[y] = sum;
        }
        for (int y = 0; y < ny; y++) I[(size_t) y * nx + k] = line[y];
    }
    free(line);
    free(B);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* src/bicubic_interpolation.cpp:24-39 */
static inline int clamp_flag(int x, int n, int *out)
{
    if (x < 0)  { *out = 1; return 0; }
    if (x >= n) { *out = 1; return n - 1; }
    return x;
}

/* src/bicubic_interpolation.cpp:108-123  Keys cubic (a = -1/2) in the reference's Horner form */
static inline double cubic_cell(double v0, double v1, double v2, double v3, double x)
{
    return v1 + 0.5 * x * (v2 - v0 + x * (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3
                                          + x * (3.0 * (v1 - v2) + v3 - v0)));
}

/* src/bicubic_interpolation.cpp:153-245 with BOUNDARY_CONDITION 0 (Neumann).
 * Quirks kept: truncation (int)uu toward zero (:170); tap direction follows the sign of the
 * coordinate (:162-163); `my` uses sx, not sy (:173); the fractional offset is taken against the
 * CLAMPED base index (:243); any clamped tap zeroes the sample when border_out (:214-215);
 * columns are interpolated in y first, then one cubic in x (:130-145, :236-243). */
double orc_bicubic_at(const double *in, double uu, double vv, int nx, int ny, int border_out)
{
    const int sx = (uu < 0) ? -1 : 1;
    const int sy = (vv < 0) ? -1 : 1;
    int out = 0;
    const int x   = clamp_flag((int) uu, nx, &out);
    const int y   = clamp_flag((int) vv, ny, &out);
    const int mx  = clamp_flag((int) uu - sx, nx, &out);
    const int my  = clamp_flag((int) vv - sx, ny, &out);
    const int dx  = clamp_flag((int) uu + sx, nx, &out);
    const int dy  = clamp_flag((int) vv + sy, ny, &out);
    const int ddx = clamp_flag((int) uu + 2 * sx, nx, &out);
    const int ddy = clamp_flag((int) vv + 2 * sy, ny, &out);

    if (out && border_out) return 0.0;

    const double fx = uu - x, fy = vv - y;
    const int col[4] = { mx, x, dx, ddx };
    double c[4];
    for (int k = 0; k < 4; k++)
        c[k] = cubic_cell(in[col[k] + nx * my], in[col[k] + nx * y],
                          in[col[k] + nx * dy], in[col[k] + nx * ddy], fy);
    return cubic_cell(c[0], c[1], c[2], c[3], fx);
}

/* src/bicubic_interpolation.cpp:352-374 */
void orc_bicubic_warp(const double *in, const double *u, const double *v, double *out,
                      int nx, int ny, int border_out)
{
    #pragma omp parallel for
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nx; j++) {
            const int p = i * nx + j;
            out[p] = orc_bicubic_at(in, j + u[p], i + v[p], nx, ny, border_out);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* src/zoom.cpp:22-34 */
void orc_zoom_size(int nx, int ny, int *nxx, int *nyy, double factor)
{
    *nxx = (int) (nx * factor + 0.5);
    *nyy = (int) (ny * factor + 0.5);
}

/* src/zoom.cpp:41-78 */
int orc_zoom_out(const double *I, double *Iout, int nx, int ny, double factor)
{
    double *Is = dalloc((size_t) nx * ny);
    memcpy(Is, I, (size_t) nx * ny * sizeof(double));
    int nxx, nyy;
    orc_zoom_size(nx, ny, &nxx, &nyy, factor);
    const double sigma = ZOOM_SIGMA_ZERO * sqrt(1.0 / (factor * factor) - 1.0);
    if (orc_gaussian(Is, nx, ny, sigma)) { free(Is); return 1; }
    #pragma omp parallel for
    for (int i1 = 0; i1 < nyy; i1++)
        for (int j1 = 0; j1 < nxx; j1++) {
            const double i2 = i1 / factor, j2 = j1 / factor;
            Iout[i1 * nxx + j1] = orc_bicubic_at(Is, j2, i2, nx, ny, 0);
        }
    free(Is);
    return 0;
}

/* src/zoom.cpp:132-155 */
void orc_zoom_in(const double *I, double *Iout, int nx, int ny, int nxx, int nyy)
{
    const double factorx = ((double) nxx / nx);
    const double factory = ((double) nyy / ny);
    #pragma omp parallel for
    for (int i1 = 0; i1 < nyy; i1++)
        for (int j1 = 0; j1 < nxx; j1++) {
            const double i2 = i1 / factory, j2 = j1 / factorx;
            Iout[i1 * nxx + j1] = orc_bicubic_at(I, j2, i2, nx, ny, 0);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* src/utils.cpp:283-326 + getminmax :509-525 */
void orc_image_normalization_2(const double *I1, const double *I2, double *I1n, double *I2n, int size)
{
    double lo = I1[0], hi = I1[0];
    for (int i = 1; i < size; i++) { if (I1[i] < lo) lo = I1[i]; if (I1[i] > hi) hi = I1[i]; }
    double lo2 = I2[0], hi2 = I2[0];
    for (int i = 1; i < size; i++) { if (I2[i] < lo2) lo2 = I2[i]; if (I2[i] > hi2) hi2 = I2[i]; }
    if (hi2 > hi) hi = hi2;
    if (lo2 < lo) lo = lo2;
    const double den = hi - lo;
    if (den > 0) {
        #pragma omp parallel for
        for (int i = 0; i < size; i++) {
            I1n[i] = 255.0 * (I1[i] - lo) / den;
            I2n[i] = 255.0 * (I2[i] - lo) / den;
        }
    } else {
        #pragma omp parallel for
        for (int i = 0; i < size; i++) { I1n[i] = I1[i]; I2n[i] = I2[i]; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* TV-L1 inner iterations, src/tvl1flow.cpp:113-182: five separate sweeps per iteration exactly as
 * the reference runs them (threshold -> div p -> u update + error -> grad u -> dual update). */
typedef struct {
    double *v1, *v2, *div_p1, *div_p2, *u1x, *u1y, *u2x, *u2y;
} tvl1_scratch;

static void tvl1_scratch_alloc(tvl1_scratch *s, size_t n)
{
    s->v1 = dalloc(n); s->v2 = dalloc(n); s->div_p1 = dalloc(n); s->div_p2 = dalloc(n);
    s->u1x = dalloc(n); s->u1y = dalloc(n); s->u2x = dalloc(n); s->u2y = dalloc(n);
}

static void tvl1_scratch_free(tvl1_scratch *s)
{
    free(s->v1); free(s->v2); free(s->div_p1); free(s->div_p2);
    free(s->u1x); free(s->u1y); free(s->u2x); free(s->u2y);
}

static double tvl1_one_iteration(double *u1, double *u2, double *p11, double *p12, double *p21,
                                 double *p22, const double *I1wx, const double *I1wy,
                                 const double *rho_c, const double *grad, tvl1_scratch *s,
                                 int nx, int ny, double tau, double theta, double l_t)
{
    const int size = nx * ny;
    double *v1 = s->v1, *v2 = s->v2;

    /* thresholding operator TH, :117-143 */
    #pragma omp parallel for
    for (int i = 0; i < size; i++) {
        const double rho = rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
        double d1, d2;
        if (rho < -l_t * grad[i]) {
            d1 = l_t * I1wx[i];
            d2 = l_t * I1wy[i];
        } else if (rho > l_t * grad[i]) {
            d1 = -l_t * I1wx[i];
            d2 = -l_t * I1wy[i];
        } else if (grad[i] < TVL1_GRAD_IS_ZERO) {
            d1 = d2 = 0;
        } else {
            const double fi = -rho / grad[i];
            d1 = fi * I1wx[i];
            d2 = fi * I1wy[i];
        }
        v1[i] = u1[i] + d1;
        v2[i] = u2[i] + d2;
    }

    orc_divergence(p11, p12, s->div_p1, nx, ny);      /* :146 */
    orc_divergence(p21, p22, s->div_p2, nx, ny);      /* :147 */

    /* primal update + convergence error, :150-162 */
    double error = 0.0;
    #pragma omp parallel for reduction(+:error)
    for (int i = 0; i < size; i++) {
        const double u1k = u1[i], u2k = u2[i];
        u1[i] = v1[i] + theta * s->div_p1[i];
        u2[i] = v2[i] + theta * s->div_p2[i];
        error += (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
    }
    error /= size;

    orc_forward_gradient(u1, s->u1x, s->u1y, nx, ny); /* :165 */
    orc_forward_gradient(u2, s->u2x, s->u2y, nx, ny); /* :166 */

    /* dual update, :169-181 */
    #pragma omp parallel for
    for (int i = 0; i < size; i++) {
        const double taut = tau / theta;
        const double g1 = hypot(s->u1x[i], s->u1y[i]);
        const double g2 = hypot(s->u2x[i], s->u2y[i]);
        const double ng1 = 1.0 + taut * g1;
        const double ng2 = 1.0 + taut * g2;
        p11[i] = (p11[i] + taut * s->u1x[i]) / ng1;
        p12[i] = (p12[i] + taut * s->u1y[i]) / ng1;
        p21[i] = (p21[i] + taut * s->u2x[i]) / ng2;
        p22[i] = (p22[i] + taut * s->u2y[i]) / ng2;
    }
    return error;
}

double orc_tvl1_iterations(double *u1, double *u2, double *p11, double *p12, double *p21, double *p22,
                           const double *I1wx, const double *I1wy, const double *rho_c,
                           const double *grad, int nx, int ny, double tau, double lambda,
                           double theta, int n_iter)
{
    tvl1_scratch s;
    tvl1_scratch_alloc(&s, (size_t) nx * ny);
    double error = INFINITY;
    for (int n = 0; n < n_iter; n++)
        error = tvl1_one_iteration(u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, grad, &s,
                                   nx, ny, tau, theta, lambda * theta);
    tvl1_scratch_free(&s);
    return error;
}

/* src/tvl1flow.cpp:46-212 */
void orc_tvl1_single_scale(const double *I0, const double *I1, double *u1, double *u2, int nx, int ny,
                           double tau, double lambda, double theta, int warps, double epsilon,
                           int verbose, int *iters, double *errs)
{
    const int size = nx * ny;
    const size_t n = (size_t) size;
    const double l_t = lambda * theta;
    double *I1x = dalloc(n), *I1y = dalloc(n), *I1w = dalloc(n), *I1wx = dalloc(n), *I1wy = dalloc(n);
    double *rho_c = dalloc(n), *grad = dalloc(n);
    double *p11 = dalloc(n), *p12 = dalloc(n), *p21 = dalloc(n), *p22 = dalloc(n);
    tvl1_scratch s;
    tvl1_scratch_alloc(&s, n);

    orc_centered_gradient(I1, I1x, I1y, nx, ny);                       /* :84 */
    for (int i = 0; i < size; i++) p11[i] = p12[i] = p21[i] = p22[i] = 0.0;   /* :87-90 */

    for (int w = 0; w < warps; w++) {
        orc_bicubic_warp(I1,  u1, u2, I1w,  nx, ny, 1);                /* :94-96 */
        orc_bicubic_warp(I1x, u1, u2, I1wx, nx, ny, 1);
        orc_bicubic_warp(I1y, u1, u2, I1wy, nx, ny, 1);

        #pragma omp parallel for
        for (int i = 0; i < size; i++) {                               /* :98-109 */
            const double Ix2 = I1wx[i] * I1wx[i];
            const double Iy2 = I1wy[i] * I1wy[i];
            grad[i] = (Ix2 + Iy2);
            rho_c[i] = (I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i]);
        }

        int it = 0;
        double error = INFINITY;
        while (error > epsilon * epsilon && it < TVL1_MAX_ITERATIONS) {  /* :113 */
            it++;
            error = tvl1_one_iteration(u1, u2, p11, p12, p21, p22, I1wx, I1wy, rho_c, grad, &s,
                                       nx, ny, tau, theta, l_t);
        }
        if (verbose)
            fprintf(stderr, "Warping: %d, Iterations: %d, Error: %f\n", w, it, error);  /* :184-188 */
        if (iters) iters[w] = it;
        if (errs) errs[w] = error;
    }

    tvl1_scratch_free(&s);
    free(I1x); free(I1y); free(I1w); free(I1wx); free(I1wy); free(rho_c); free(grad);
    free(p11); free(p12); free(p21); free(p22);
}

/* Shared pyramid prologue: src/tvl1flow.cpp:236-280, horn_schunck_pyramidal.cpp:279-323,
 * brox_optic_flow_spatial.cpp:467-509 are the same code. */
typedef struct {
    int nscales;
    int *nx, *ny;
    double **A, **B, **u, **v;
} pyramid;

static int pyramid_build(pyramid *P, const double *Ia, const double *Ib, double *u, double *v,
                         int nx, int ny, int nscales, double zfactor, double presmooth)
{
    P->nscales = nscales;
    P->nx = (int *) malloc(sizeof(int) * nscales);
    P->ny = (int *) malloc(sizeof(int) * nscales);
    P->A = (double **) calloc(nscales, sizeof(double *));
    P->B = (double **) calloc(nscales, sizeof(double *));
    P->u = (double **) calloc(nscales, sizeof(double *));
    P->v = (double **) calloc(nscales, sizeof(double *));
    P->nx[0] = nx; P->ny[0] = ny;
    P->A[0] = dalloc((size_t) nx * ny);
    P->B[0] = dalloc((size_t) nx * ny);
    P->u[0] = u; P->v[0] = v;
    int rc = 0;
    orc_image_normalization_2(Ia, Ib, P->A[0], P->B[0], nx * ny);
    rc |= orc_gaussian(P->A[0], nx, ny, presmooth);
    rc |= orc_gaussian(P->B[0], nx, ny, presmooth);
    for (int s = 1; s < nscales && !rc; s++) {
        orc_zoom_size(P->nx[s - 1], P->ny[s - 1], &P->nx[s], &P->ny[s], zfactor);
        const size_t n = (size_t) P->nx[s] * P->ny[s];
        P->A[s] = dalloc(n); P->B[s] = dalloc(n); P->u[s] = dalloc(n); P->v[s] = dalloc(n);
        rc |= orc_zoom_out(P->A[s - 1], P->A[s], P->nx[s - 1], P->ny[s - 1], zfactor);
        rc |= orc_zoom_out(P->B[s - 1], P->B[s], P->nx[s - 1], P->ny[s - 1], zfactor);
    }
    if (!rc) {
        const int c = nscales - 1;
        for (int i = 0; i < P->nx[c] * P->ny[c]; i++) P->u[c][i] = P->v[c][i] = 0.0;
    }
    return rc;
}

/* zoom the flow to the next finer level and rescale it: tvl1flow.cpp:302-309 */
static void pyramid_upsample(pyramid *P, int s, double zfactor)
{
    orc_zoom_in(P->u[s], P->u[s - 1], P->nx[s], P->ny[s], P->nx[s - 1], P->ny[s - 1]);
    orc_zoom_in(P->v[s], P->v[s - 1], P->nx[s], P->ny[s], P->nx[s - 1], P->ny[s - 1]);
    const int n = P->nx[s - 1] * P->ny[s - 1];
    for (int i = 0; i < n; i++) {
        P->u[s - 1][i] *= 1.0 / zfactor;
        P->v[s - 1][i] *= 1.0 / zfactor;
    }
}

static void pyramid_free(pyramid *P)
{
    for (int s = 0; s < P->nscales; s++) {
        free(P->A[s]); free(P->B[s]);
        if (s) { free(P->u[s]); free(P->v[s]); }
    }
    free(P->A); free(P->B); free(P->u); free(P->v); free(P->nx); free(P->ny);
}

/* src/tvl1flow.cpp:219-328 */
int orc_tvl1_multiscale(const double *I0, const double *I1, double *u1, double *u2, int nx, int ny,
                        double tau, double lambda, double theta, int nscales, double zfactor,
                        int warps, double epsilon, int verbose, int *iters, double *errs)
{
    pyramid P;
    int rc = pyramid_build(&P, I0, I1, u1, u2, nx, ny, nscales, zfactor, TVL1_PRESMOOTH_SIGMA);
    for (int s = nscales - 1; s >= 0 && !rc; s--) {
        if (verbose) fprintf(stderr, "Scale %d: %dx%d\n", s, P.nx[s], P.ny[s]);
        orc_tvl1_single_scale(P.A[s], P.B[s], P.u[s], P.v[s], P.nx[s], P.ny[s], tau, lambda, theta,
                              warps, epsilon, verbose, iters ? iters + s * warps : NULL,
                              errs ? errs + s * warps : NULL);
        if (s) pyramid_upsample(&P, s, zfactor);
    }
    pyramid_free(&P);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Horn-Schunck SOR point update, src/horn_schunck_pyramidal.cpp:31-71.  nb[0..3] = diagonal
 * neighbours (weight 1/12), nb[4..7] = axial neighbours (weight 1/6), in the reference order
 * p1..p8; v is updated with the NEW u (:66-67). */
static inline double hs_sor_point(const double *Au, const double *Av, const double *Du,
                                  const double *Dv, const double *D, double *u, double *v,
                                  double al, int p, int p1, int p2, int p3, int p4, int p5, int p6,
                                  int p7, int p8)
{
    const double w = HS_SOR_W;
    const double ula = 1. / 12. * (u[p1] + u[p2] + u[p3] + u[p4]) + 1. / 6. * (u[p5] + u[p6] + u[p7] + u[p8]);
    const double vla = 1. / 12. * (v[p1] + v[p2] + v[p3] + v[p4]) + 1. / 6. * (v[p5] + v[p6] + v[p7] + v[p8]);
    const double uk = u[p], vk = v[p];
    u[p] = (1.0 - w) * uk + w * (Au[p] - D[p] * v[p] + al * ula) / Du[p];
    v[p] = (1.0 - w) * vk + w * (Av[p] - D[p] * u[p] + al * vla) / Dv[p];
    return (u[p] - uk) * (u[p] - uk) + (v[p] - vk) * (v[p] - vk);
}

/* One full sweep in the reference's visiting order (:143-231): interior rows lexicographic (under
 * the reference's racy `omp parallel for`, deterministic with one thread), then first/last row
 * interleaved per column, first/last column interleaved per row, then the four corners.  Border
 * pixels replace missing neighbours by replicated indices exactly as :161-228 pass them. */
/* The reference's replicated border indices are exactly the clamped neighbour coordinates, in the
 * order up-left, up-right, bottom-left, bottom-right / up, left, bottom, right -- with ONE exception:
 * the bottom-right corner passes its four diagonal taps as (k-1, k, k-nx-1, k-nx)
 * (horn_schunck_pyramidal.cpp:222-228), i.e. the bottom pair first.  Same set, different summation
 * order, so it is kept to stay bit-identical. */
static inline double hs_point_clamped(const double *Au, const double *Av, const double *Du, const double *Dv,
                                      const double *D, double *u, double *v, double a2, int i, int j, int nx, int ny)
{
    const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
    const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
    if (i == ny - 1 && j == nx - 1)
        return hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, i * nx + j, id * nx + jl, id * nx + jr, iu * nx + jl,
                            iu * nx + jr, iu * nx + j, i * nx + jl, id * nx + j, i * nx + jr);
    return hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, i * nx + j, iu * nx + jl, iu * nx + jr, id * nx + jl,
                        id * nx + jr, iu * nx + j, i * nx + jl, id * nx + j, i * nx + jr);
}

/* Checker aid (NOT reference behaviour): sweep order used by the GPU kernels.  0 = the reference's
 * lexicographic order (default); 1 = multi-colour order -- Horn-Schunck: 4 colours (i%2, j%2) in the
 * order (0,0) (0,1) (1,0) (1,1); Brox: red-black, (i+j)%2 == 0 first -- every pixel (borders included)
 * with clamped neighbour indices, which is what the reference's replicated border indices amount to.
 * With order 1 the oracle reproduces the HIP path's arithmetic bit for bit, which separates "is the
 * kernel right" from "how far does the colouring move the result" (tests/test_gpu_sor.py). */
static int g_sor_order = 0;
void orc_set_sor_order(int order) { g_sor_order = order; }
/* Checker aid for the HIP path's option "sor_colour_levels": with order 1, pyramid level s is swept in colour order only
 * when bit s of the mask is set; the other levels keep the reference's order (0). */
static unsigned g_sor_colour_levels = ~0u;
void orc_set_sor_colour_levels(unsigned mask) { g_sor_colour_levels = mask; }
/* ... and "sor_exact_tail": with order 1, the last `tail` solves of the finest level (level 0) keep the reference's order */
static int g_sor_exact_tail = 0, g_sor_level = -1;
void orc_set_sor_exact_tail(int tail) { g_sor_exact_tail = tail < 0 ? 0 : tail; }
static int sor_order_enter_level(int s)
{
    const int saved = g_sor_order;
    g_sor_level = s;
    if (g_sor_order == 1 && s < 32 && !((g_sor_colour_levels >> s) & 1u)) g_sor_order = 0;
    return saved;
}
/* order 3 (Brox; checker aid for the HIP path's tolerance mode, ofx_sor_tile.hip k_brox_wave): the image cut into tiles of
 * g_tile_w x g_tile_h pixels coloured as a checkerboard; a sweep updates the tiles of colour 0, then those of colour 1, each
 * tile in row-major order, in place.  With order 1, the Brox levels below g_sor_wave_levels use it (the HIP option
 * "sor_wave_levels"; 0 = none, the default here). */
static int g_tile_w = 64, g_tile_h = 64, g_sor_wave_levels = 0;
void orc_set_sor_tile(int w, int h) { g_tile_w = w; g_tile_h = h; }
void orc_set_sor_wave_levels(int n) { g_sor_wave_levels = n < 0 ? 0 : n; }
/* order of solve `k` of `n` at the current level */
static int sor_order_of_solve(int k, int n)
{
    if (g_sor_order == 1 && g_sor_level == 0 && k >= n - g_sor_exact_tail) return 0;
    return g_sor_order;
}

static double hs_sweep_coloured(const double *Au, const double *Av, const double *Du, const double *Dv,
                                const double *D, double *u, double *v, double a2, int nx, int ny)
{
    double error = 0;
    for (int col = 0; col < 4; col++) {
        const int ci = col >> 1, cj = col & 1;
        for (int i = ci; i < ny; i += 2)
            for (int j = cj; j < nx; j += 2)
                error += hs_point_clamped(Au, Av, Du, Dv, D, u, v, a2, i, j, nx, ny);
    }
    return error;
}

/* Checker aid, order 2: the HIP path's EXACT schedule.  Pixel X of sweep s runs at time
 * T = pos(X) + C*s; everything with the same T runs "at once" (here: in arbitrary order).  pos/C are
 * chosen so that for every pair of neighbouring pixels X before Y in the reference's sweep order
 * T(X,s) < T(Y,s) < T(X,s+1): each pixel then reads exactly the versions the sequential sweep reads,
 * although up to `nsweeps` sweeps are in flight.  Derivation: DESIGN.md 5.3. */
static int hs_pos(int i, int j, int nx, int ny)
{
    const int top = (i == 0), bot = (i == ny - 1), lef = (j == 0), rig = (j == nx - 1);
    if (!top && !bot && !lef && !rig) return 2 * i + j;      /* interior, lexicographic */
    if (top && !lef && !rig) return j + 4;                    /* first row, after the interior */
    if (bot && !lef && !rig) return 2 * (ny - 1) + j;         /* last row */
    if (lef && !top && !bot) return 2 * i + 4;                /* first column, after the rows */
    if (rig && !top && !bot) return 2 * i + nx + 1;           /* last column */
    if (top && lef) return 7;                                 /* corners last */
    if (top && rig) return nx + 4;
    if (bot && lef) return 2 * ny + 1;
    return 2 * ny + nx - 2;
}
#define HS_PLANE_C 6

static int brox_pos(int i, int j, int nx, int ny)
{
    const int top = (i == 0), bot = (i == ny - 1), lef = (j == 0), rig = (j == nx - 1);
    if (!top && !bot && !lef && !rig) return i + j;
    if (top && !lef && !rig) return j + 2;
    if (bot && !lef && !rig) return ny - 1 + j;
    if (lef && !top && !bot) return i + 2;
    if (rig && !top && !bot) return i + nx - 1;
    if (top && lef) return 4;
    if (top && rig) return nx + 1;
    if (bot && lef) return ny + 1;
    return ny + nx - 2;
}
#define BROX_PLANE_C 2

/* pixels bucketed by pos: start[p] .. start[p+1] index into `order` */
typedef struct { int npos; int *start; int *order; } plane_index;

static void plane_index_build(plane_index *P, int nx, int ny, int (*pos)(int, int, int, int))
{
    int maxp = 0;
    for (int i = 0; i < ny; i++) for (int j = 0; j < nx; j++) { int q = pos(i, j, nx, ny); if (q > maxp) maxp = q; }
    P->npos = maxp + 1;
    P->start = (int *) calloc((size_t) P->npos + 1, sizeof(int));
    P->order = (int *) malloc(sizeof(int) * (size_t) nx * ny);
    for (int i = 0; i < ny; i++) for (int j = 0; j < nx; j++) P->start[pos(i, j, nx, ny) + 1]++;
    for (int q = 0; q < P->npos; q++) P->start[q + 1] += P->start[q];
    int *fill = (int *) malloc(sizeof(int) * (size_t) P->npos);
    memcpy(fill, P->start, sizeof(int) * (size_t) P->npos);
    for (int i = 0; i < ny; i++) for (int j = 0; j < nx; j++) P->order[fill[pos(i, j, nx, ny)]++] = i * nx + j;
    free(fill);
}

static void plane_index_free(plane_index *P) { free(P->start); free(P->order); }

/* runs sweeps [0, nsweeps) pipelined; errs[s] = sum of squared updates of sweep s */
static void hs_sweeps_planes(const plane_index *P, const double *Au, const double *Av, const double *Du,
                             const double *Dv, const double *D, double *u, double *v, double a2, int nx, int ny,
                             int nsweeps, double *errs)
{
    for (int s = 0; s < nsweeps; s++) errs[s] = 0;
    const int tmax = P->npos - 1 + HS_PLANE_C * (nsweeps - 1);
    for (int t = 0; t <= tmax; t++)
        for (int s = 0; s < nsweeps; s++) {
            const int q = t - HS_PLANE_C * s;
            if (q < 0 || q >= P->npos) continue;
            for (int e = P->start[q]; e < P->start[q + 1]; e++) {
                const int k = P->order[e];
                errs[s] += hs_point_clamped(Au, Av, Du, Dv, D, u, v, a2, k / nx, k % nx, nx, ny);
            }
        }
}

static int g_plane_batch = 64;      /* sweeps in flight per batch, like the HIP path */
void orc_set_plane_batch(int b) { if (b > 0) g_plane_batch = b; }

static double hs_sweep(const double *Au, const double *Av, const double *Du, const double *Dv,
                       const double *D, double *u, double *v, double a2, int nx, int ny)
{
    if (g_sor_order == 1) return hs_sweep_coloured(Au, Av, Du, Dv, D, u, v, a2, nx, ny);
    double error = 0;
    #pragma omp parallel for reduction(+:error)
    for (int i = 1; i < ny - 1; i++)
        for (int j = 1; j < nx - 1; j++) {
            const int k = i * nx + j;
            error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - nx - 1, k - nx + 1,
                                  k + nx - 1, k + nx + 1, k - nx, k - 1, k + nx, k + 1);
        }
    for (int j = 1; j < nx - 1; j++) {
        int k = j;
        error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - 1, k + 1, k + nx - 1, k + nx + 1,
                              k, k - 1, k + nx, k + 1);
        k = (ny - 1) * nx + j;
        error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - nx - 1, k - nx + 1, k - 1, k + 1,
                              k - nx, k - 1, k, k + 1);
    }
    for (int i = 1; i < ny - 1; i++) {
        int k = i * nx;
        error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - nx, k - nx + 1, k + nx, k + nx + 1,
                              k - nx, k, k + nx, k + 1);
        k = (i + 1) * nx - 1;
        error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - nx - 1, k - nx, k + nx - 1, k + nx,
                              k - nx, k - 1, k + nx, k);
    }
    error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, 0, 0, 1, nx, nx + 1, 0, 0, nx, 1);
    int k = nx - 1;
    error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - 1, k, k + nx - 1, k + nx, k, k - 1, k + nx, k);
    k = (ny - 1) * nx;
    error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - nx, k - nx + 1, k, k + 1, k - nx, k, k, k + 1);
    k = ny * nx - 1;
    error += hs_sor_point(Au, Av, Du, Dv, D, u, v, a2, k, k - 1, k, k - nx - 1, k - nx, k - nx, k - 1, k, k);
    return error;
}

/* src/horn_schunck_pyramidal.cpp:78-249 */
void orc_hs_single_scale(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                         double alpha, int warps, double TOL, int maxiter, int verbose, int *iters)
{
    const int size = nx * ny;
    const size_t n = (size_t) size;
    const double alpha2 = alpha * alpha;
    double *I2x = dalloc(n), *I2y = dalloc(n), *I2w = dalloc(n), *I2wx = dalloc(n), *I2wy = dalloc(n);
    double *Au = dalloc(n), *Av = dalloc(n), *Du = dalloc(n), *Dv = dalloc(n), *D = dalloc(n);

    if (verbose)
        fprintf(stderr, "Single-scale Horn-Schunck of a %dx%d image\n\ta=%g nw=%d eps=%g mi=%d v=%d\n",
                nx, ny, alpha, warps, TOL, maxiter, verbose);

    orc_centered_gradient(I2, I2x, I2y, nx, ny);                       /* :114 */
    for (int w = 0; w < warps; w++) {
        if (verbose) fprintf(stderr, "Warping %d:", w);
        orc_bicubic_warp(I2,  u, v, I2w,  nx, ny, 1);                  /* :123-125 */
        orc_bicubic_warp(I2x, u, v, I2wx, nx, ny, 1);
        orc_bicubic_warp(I2y, u, v, I2wy, nx, ny, 1);
        for (int i = 0; i < size; i++) {                               /* :128-137 */
            const double I2wl = I2wx[i] * u[i] + I2wy[i] * v[i];
            const double dif = I1[i] - I2w[i] + I2wl;
            Au[i] = dif * I2wx[i];
            Av[i] = dif * I2wy[i];
            Du[i] = I2wx[i] * I2wx[i] + alpha2;
            Dv[i] = I2wy[i] * I2wy[i] + alpha2;
            D[i]  = I2wx[i] * I2wy[i];
        }
        int niter = 0;
        double error = 1000;
        const int order_all = g_sor_order;
        g_sor_order = sor_order_of_solve(w, warps);
        if (g_sor_order == 2 && nx >= 3 && ny >= 3) {
            /* batches of pipelined sweeps; a batch that runs past the stopping sweep is rolled back to
             * its checkpoint and re-run with exactly the sweeps that count */
            plane_index PI;
            plane_index_build(&PI, nx, ny, hs_pos);
            double *cu = dalloc(n), *cv = dalloc(n), *errs = dalloc((size_t) g_plane_batch);
            while (error > TOL && niter < maxiter) {
                const int b = (maxiter - niter < g_plane_batch) ? maxiter - niter : g_plane_batch;
                memcpy(cu, u, n * sizeof(double));
                memcpy(cv, v, n * sizeof(double));
                hs_sweeps_planes(&PI, Au, Av, Du, Dv, D, u, v, alpha2, nx, ny, b, errs);
                int used = b;
                for (int q = 0; q < b; q++) { error = sqrt(errs[q] / size); if (!(error > TOL)) { used = q + 1; break; } }
                if (used < b) {
                    memcpy(u, cu, n * sizeof(double));
                    memcpy(v, cv, n * sizeof(double));
                    hs_sweeps_planes(&PI, Au, Av, Du, Dv, D, u, v, alpha2, nx, ny, used, errs);
                    error = sqrt(errs[used - 1] / size);
                }
                niter += used;
            }
            free(cu); free(cv); free(errs);
            plane_index_free(&PI);
        } else
        while (error > TOL && niter < maxiter) {                       /* :143 */
            niter++;
            error = hs_sweep(Au, Av, Du, Dv, D, u, v, alpha2, nx, ny);
            error = sqrt(error / size);                                /* :230 */
        }
        g_sor_order = order_all;
        if (verbose) fprintf(stderr, "Iterations %d (%g)\n", niter, error);
        if (iters) iters[w] = niter;
    }
    free(I2x); free(I2y); free(I2w); free(I2wx); free(I2wy);
    free(Au); free(Av); free(Du); free(Dv); free(D);
}

/* src/horn_schunck_pyramidal.cpp:258-370 */
int orc_hs_pyramidal(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                     double alpha, int nscales, double zfactor, int warps, double TOL, int maxiter,
                     int verbose, int *iters)
{
    if (verbose)
        fprintf(stderr, "Multiscale Horn-Schunck of a %dx%d pair\n\ta=%g ns=%d zf=%g nw=%d eps=%g mi=%d\n",
                nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter);
    pyramid P;
    int rc = pyramid_build(&P, I1, I2, u, v, nx, ny, nscales, zfactor, HS_PRESMOOTH_SIGMA);
    for (int s = nscales - 1; s >= 0 && !rc; s--) {
        if (verbose) fprintf(stderr, "Scale: %d %dx%d\n", s, P.nx[s], P.ny[s]);
        const int order = sor_order_enter_level(s);
        orc_hs_single_scale(P.A[s], P.B[s], P.u[s], P.v[s], P.nx[s], P.ny[s], alpha, warps, TOL,
                            maxiter, verbose, iters ? iters + s * warps : NULL);
        g_sor_order = order;
        if (s) pyramid_upsample(&P, s, zfactor);
    }
    pyramid_free(&P);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Brox spatial.  src/brox_spatial_mask.cpp:16-93: psi1..4 = half-sums with the down / up / right /
 * left neighbour, 0 across the image border. */
static void brox_psi_divergence(const double *psi, double *psi1, double *psi2, double *psi3,
                                double *psi4, int nx, int ny)
{
    #pragma omp parallel for
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nx; j++) {
            const int k = i * nx + j;
            psi1[k] = (i < ny - 1) ? 0.5 * (psi[k + nx] + psi[k]) : 0;
            psi2[k] = (i > 0)      ? 0.5 * (psi[k - nx] + psi[k]) : 0;
            psi3[k] = (j < nx - 1) ? 0.5 * (psi[k + 1] + psi[k]) : 0;
            psi4[k] = (j > 0)      ? 0.5 * (psi[k - 1] + psi[k]) : 0;
        }
}

/* src/brox_spatial_mask.cpp:100-171: sum of psi_k * (neighbour - centre) over the neighbours that
 * exist, accumulated in the order down, up, right, left (missing terms are left out of the
 * expression, not added as zeros). */
static inline double brox_div_at(const double *f, const double *psi1, const double *psi2,
                                 const double *psi3, const double *psi4, int i, int j, int nx, int ny)
{
    const int k = i * nx + j;
    double acc = 0;
    int have = 0;
    if (i < ny - 1) { acc = psi1[k] * (f[k + nx] - f[k]); have = 1; }
    if (i > 0)      { const double t = psi2[k] * (f[k - nx] - f[k]); acc = have ? acc + t : t; have = 1; }
    if (j < nx - 1) { const double t = psi3[k] * (f[k + 1] - f[k]);  acc = have ? acc + t : t; have = 1; }
    if (j > 0)      { const double t = psi4[k] * (f[k - 1] - f[k]);  acc = have ? acc + t : t; have = 1; }
    return acc;
}

static void brox_divergence_u(const double *u, const double *v, const double *psi1,
                              const double *psi2, const double *psi3, const double *psi4,
                              double *div_u, double *div_v, int nx, int ny)
{
    #pragma omp parallel for
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nx; j++) {
            div_u[i * nx + j] = brox_div_at(u, psi1, psi2, psi3, psi4, i, j, nx, ny);
            div_v[i * nx + j] = brox_div_at(v, psi1, psi2, psi3, psi4, i, j, nx, ny);
        }
}

/* src/brox_optic_flow_spatial.cpp:129-172.  (i0, i1, j0, j1) are the OFFSETS to the previous /
 * following row and column; the reference passes 0 for a missing neighbour so the tap lands on
 * the pixel itself (:332-388), with psi = 0 there. */
static inline double brox_sor_point(const double *Au, const double *Av, const double *Du,
                                    const double *Dv, const double *D, double *du, double *dv,
                                    double alpha, const double *psi1, const double *psi2,
                                    const double *psi3, const double *psi4, int i, int i0, int i1,
                                    int j, int nx, int j0, int j1)
{
    const double w = BROX_SOR_W;
    const int k = i * nx + j;
    const double div_du = psi1[k] * du[k + i1] + psi2[k] * du[k - i0] + psi3[k] * du[k + j1] + psi4[k] * du[k - j0];
    const double div_dv = psi1[k] * dv[k + i1] + psi2[k] * dv[k - i0] + psi3[k] * dv[k + j1] + psi4[k] * dv[k - j0];
    const double duk = du[k], dvk = dv[k];
    du[k] = (1. - w) * du[k] + w * (Au[k] - D[k] * dv[k] + alpha * div_du) / Du[k];
    dv[k] = (1. - w) * dv[k] + w * (Av[k] - D[k] * du[k] + alpha * div_dv) / Dv[k];
    return (du[k] - duk) * (du[k] - duk) + (dv[k] - dvk) * (dv[k] - dvk);
}

/* src/brox_optic_flow_spatial.cpp:179-444 */
static void brox_single_scale(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                              double alpha, double gamma, double TOL, int inner_iter, int outer_iter,
                              int nthreads, int verbose, int *iters)
{
    const int size = nx * ny;
    const size_t n = (size_t) size;
    enum { NARR = 34 };
    double *a[NARR];
    for (int q = 0; q < NARR; q++) a[q] = dalloc(n);
    double *du = a[0], *dv = a[1], *ux = a[2], *uy = a[3], *vx = a[4], *vy = a[5];
    double *I1x = a[6], *I1y = a[7], *I2x = a[8], *I2y = a[9], *I2w = a[10], *I2wx = a[11], *I2wy = a[12];
    double *I2xx = a[13], *I2yy = a[14], *I2xy = a[15], *I2wxx = a[16], *I2wyy = a[17], *I2wxy = a[18];
    double *div_u = a[19], *div_v = a[20], *div_d = a[21];
    double *Au = a[22], *Av = a[23], *Du = a[24], *Dv = a[25], *D = a[26];
    double *psid = a[27], *psig = a[28], *psis = a[29], *psi1 = a[30], *psi2 = a[31], *psi3 = a[32], *psi4 = a[33];
    int solve = 0;
    (void) nthreads;

    orc_centered_gradient(I1, I1x, I1y, nx, ny);               /* :235-236 */
    orc_centered_gradient(I2, I2x, I2y, nx, ny);
    orc_dxx(I2, I2xx, nx, ny);                                 /* :239-241 */
    orc_dyy(I2, I2yy, nx, ny);
    orc_dxy(I2, I2xy, nx, ny);

    for (int no = 0; no < outer_iter; no++) {                  /* :244 */
        orc_bicubic_warp(I2,   u, v, I2w,   nx, ny, 1);        /* :246-251 */
        orc_bicubic_warp(I2x,  u, v, I2wx,  nx, ny, 1);
        orc_bicubic_warp(I2y,  u, v, I2wy,  nx, ny, 1);
        orc_bicubic_warp(I2xx, u, v, I2wxx, nx, ny, 1);
        orc_bicubic_warp(I2xy, u, v, I2wxy, nx, ny, 1);
        orc_bicubic_warp(I2yy, u, v, I2wyy, nx, ny, 1);

        orc_centered_gradient(u, ux, uy, nx, ny);              /* :254-255 */
        orc_centered_gradient(v, vx, vy, nx, ny);

        #pragma omp parallel for
        for (int i = 0; i < size; i++) {                       /* psi_smooth :99-122 */
            const double gu = ux[i] * ux[i] + uy[i] * uy[i];
            const double gv = vx[i] * vx[i] + vy[i] * vy[i];
            const double d2 = gu + gv;
            psis[i] = 1. / sqrt(d2 + BROX_EPSILON * BROX_EPSILON);
        }
        brox_psi_divergence(psis, psi1, psi2, psi3, psi4, nx, ny);                   /* :261 */
        brox_divergence_u(u, v, psi1, psi2, psi3, psi4, div_u, div_v, nx, ny);       /* :264 */

        #pragma omp parallel for
        for (int i = 0; i < size; i++) {                       /* :266-274 */
            div_d[i] = alpha * (psi1[i] + psi2[i] + psi3[i] + psi4[i]);
            du[i] = dv[i] = 0;
        }

        for (int ni = 0; ni < inner_iter; ni++) {              /* :277 */
            #pragma omp parallel for
            for (int i = 0; i < size; i++) {                   /* psi_data :33-57 */
                const double dI = I2w[i] - I1[i] + I2wx[i] * du[i] + I2wy[i] * dv[i];
                const double dI2 = dI * dI;
                psid[i] = 1. / sqrt(dI2 + BROX_EPSILON * BROX_EPSILON);
            }
            #pragma omp parallel for
            for (int i = 0; i < size; i++) {                   /* psi_gradient :64-92 */
                const double dIx = I2wx[i] - I1x[i] + I2wxx[i] * du[i] + I2wxy[i] * dv[i];
                const double dIy = I2wy[i] - I1y[i] + I2wxy[i] * du[i] + I2wyy[i] * dv[i];
                const double dI2 = dIx * dIx + dIy * dIy;
                psig[i] = 1. / sqrt(dI2 + BROX_EPSILON * BROX_EPSILON);
            }
            for (int i = 0; i < size; i++) {                   /* :283-309 */
                const double p = psid[i];
                const double g = gamma * psig[i];
                const double dif = I2w[i] - I1[i];
                const double BNu = -p * dif * I2wx[i];
                const double BNv = -p * dif * I2wy[i];
                const double BDu = p * I2wx[i] * I2wx[i];
                const double BDv = p * I2wy[i] * I2wy[i];
                const double dx = (I2wx[i] - I1x[i]);
                const double dy = (I2wy[i] - I1y[i]);
                const double GNu = -g * (dx * I2wxx[i] + dy * I2wxy[i]);
                const double GNv = -g * (dx * I2wxy[i] + dy * I2wyy[i]);
                const double GDu = g * (I2wxx[i] * I2wxx[i] + I2wxy[i] * I2wxy[i]);
                const double GDv = g * (I2wyy[i] * I2wyy[i] + I2wxy[i] * I2wxy[i]);
                const double DI = (I2wxx[i] + I2wyy[i]) * I2wxy[i];
                const double Duv = p * I2wy[i] * I2wx[i] + g * DI;
                Au[i] = BNu + GNu + alpha * div_u[i];
                Av[i] = BNv + GNv + alpha * div_v[i];
                Du[i] = BDu + GDu + div_d[i];
                Dv[i] = BDv + GDv + div_d[i];
                D[i] = Duv;
            }

            double error = 1000;
            int nsor = 0;
            const int order_all = g_sor_order;
            g_sor_order = sor_order_of_solve(no * inner_iter + ni, outer_iter * inner_iter);
            if (g_sor_order == 1 && g_sor_level >= 0 && g_sor_level < g_sor_wave_levels) g_sor_order = 3;
            if (g_sor_order == 2 && nx >= 3 && ny >= 3) {
                plane_index PI;
                plane_index_build(&PI, nx, ny, brox_pos);
                double *cu = dalloc(n), *cv = dalloc(n), *errs = dalloc((size_t) g_plane_batch);
                while (error > TOL && nsor < BROX_MAXITER) {
                    const int b = (BROX_MAXITER - nsor < g_plane_batch) ? BROX_MAXITER - nsor : g_plane_batch;
                    memcpy(cu, du, n * sizeof(double));
                    memcpy(cv, dv, n * sizeof(double));
                    int used = b;
                    for (int pass = 0; pass < 2; pass++) {
                        const int ns = pass ? used : b;
                        for (int q = 0; q < ns; q++) errs[q] = 0;
                        const int tmax = PI.npos - 1 + BROX_PLANE_C * (ns - 1);
                        for (int t = 0; t <= tmax; t++)
                            for (int q = 0; q < ns; q++) {
                                const int pq = t - BROX_PLANE_C * q;
                                if (pq < 0 || pq >= PI.npos) continue;
                                for (int e = PI.start[pq]; e < PI.start[pq + 1]; e++) {
                                    const int k = PI.order[e], i = k / nx, j = k % nx;
                                    errs[q] += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i,
                                                              i > 0 ? nx : 0, i < ny - 1 ? nx : 0, j, nx, j > 0 ? 1 : 0,
                                                              j < nx - 1 ? 1 : 0);
                                }
                            }
                        if (pass) { error = sqrt(errs[used - 1] / size); break; }
                        for (int q = 0; q < b; q++) { error = sqrt(errs[q] / size); if (!(error > TOL)) { used = q + 1; break; } }
                        if (used == b) break;
                        memcpy(du, cu, n * sizeof(double));
                        memcpy(dv, cv, n * sizeof(double));
                    }
                    nsor += used;
                }
                free(cu); free(cv); free(errs);
                plane_index_free(&PI);
            } else
            while (error > TOL && nsor < BROX_MAXITER) {       /* :315 */
                error = 0;
                nsor++;
                if (g_sor_order == 3) {     /* checkerboard of tiles, row-major inside a tile, in place */
                    for (int col = 0; col < 2; col++)
                        for (int ti = 0; ti < ny; ti += g_tile_h)
                            for (int tj = 0; tj < nx; tj += g_tile_w) {
                                if (((ti / g_tile_h + tj / g_tile_w) & 1) != col) continue;
                                const int i1e = ti + g_tile_h < ny ? ti + g_tile_h : ny, j1e = tj + g_tile_w < nx ? tj + g_tile_w : nx;
                                for (int i = ti; i < i1e; i++)
                                    for (int j = tj; j < j1e; j++)
                                        error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i,
                                                                i > 0 ? nx : 0, i < ny - 1 ? nx : 0, j, nx, j > 0 ? 1 : 0,
                                                                j < nx - 1 ? 1 : 0);
                            }
                    error = sqrt(error / size);
                    continue;
                }
                if (g_sor_order == 1) {     /* checker aid: red-black order of the HIP path */
                    for (int col = 0; col < 2; col++)
                        for (int i = 0; i < ny; i++)
                            for (int j = (i + col) & 1; j < nx; j += 2)
                                error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i,
                                                        i > 0 ? nx : 0, i < ny - 1 ? nx : 0, j, nx, j > 0 ? 1 : 0,
                                                        j < nx - 1 ? 1 : 0);
                    error = sqrt(error / size);
                    continue;
                }
                #pragma omp parallel for reduction(+:error)
                for (int i = 1; i < ny - 1; i++)
                    for (int j = 1; j < nx - 1; j++)
                        error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                                i, nx, nx, j, nx, 1, 1);
                for (int j = 1; j < nx - 1; j++) {
                    error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                            0, 0, nx, j, nx, 1, 1);
                    error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                            ny - 1, nx, 0, j, nx, 1, 1);
                }
                for (int i = 1; i < ny - 1; i++) {
                    error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                            i, nx, nx, 0, nx, 0, 1);
                    error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                            i, nx, nx, nx - 1, nx, 1, 0);
                }
                error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                        0, 0, nx, 0, nx, 0, 1);
                error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                        0, 0, nx, nx - 1, nx, 1, 0);
                error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                        ny - 1, nx, 0, 0, nx, 0, 1);
                error += brox_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4,
                                        ny - 1, nx, 0, nx - 1, nx, 1, 0);
                error = sqrt(error / size);                    /* :389 */
            }
            g_sor_order = order_all;
            if (verbose) printf("Iterations: %d\n", nsor);
            if (iters) iters[solve] = nsor;
            solve++;
        }
        for (int i = 0; i < size; i++) { u[i] += du[i]; v[i] += dv[i]; }   /* :398-401 */
    }
    for (int q = 0; q < NARR; q++) free(a[q]);
}

/* src/brox_optic_flow_spatial.cpp:451-549.  `iters` is laid out [scale][outer*inner]. */
int orc_brox_spatial(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                     double alpha, double gamma, int nscales, double nu, double TOL,
                     int inner_iter, int outer_iter, int verbose, int *iters)
{
    pyramid P;
    int rc = pyramid_build(&P, I1, I2, u, v, nx, ny, nscales, nu, BROX_SIGMA);
    const int nthreads = orc_max_threads();
    for (int s = nscales - 1; s >= 0 && !rc; s--) {
        if (verbose) printf("Scale: %d\n", s);
        const int order = sor_order_enter_level(s);
        brox_single_scale(P.A[s], P.B[s], P.u[s], P.v[s], P.nx[s], P.ny[s], alpha, gamma, TOL,
                          inner_iter, outer_iter, nthreads, verbose,
                          iters ? iters + s * inner_iter * outer_iter : NULL);
        g_sor_order = order;
        if (s) pyramid_upsample(&P, s, nu);
    }
    pyramid_free(&P);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* robust_expo_methods for ONE channel (nz = 1): src/robust_expo_methods.cpp:34-567,
 * src/robust_expo_smoothness.cpp:26-187, src/robust_expo_generic_tensor.cpp:18-168.  The method is Brox's with an
 * image-driven weight `expo` in the smoothness term; what differs from brox_single_scale above, line by line:
 *   - psi1..4 are the half-sums with the RIGHT / LEFT / DOWN / UP neighbour (generic_tensor.cpp:18-90) and every sum over
 *     them runs in that order (divergence :97-168, div_d methods.cpp:266, the SOR point :138-148);
 *   - psi_smooth weighs the flow gradient with expo (smoothness.cpp:26-44); psi_data / psi_gradient add the motion terms
 *     before subtracting I1 (methods.cpp:52-56, :92-96) and accumulate onto 0;
 *   - the constant parts of the scheme are per-channel sums accumulated onto 0 and multiplied by psi afterwards
 *     (methods.cpp:283-322): -psid * (dif * I2wx), not (-psid * dif) * I2wx;
 *   - the multiscale driver presmooths with gaussian(I, nx, ny, nzz, GAUSSIAN_SIGMA) (:497-498), i.e. sigma = the number of
 *     channels (1.0) and boundary condition (int) 0.8 = 0 = DIRICHLET, and truncates alpha * nzz to an int (:529).
 * For nz > 1 the reference's pyramid reads beyond its scratch copy (zoom.cpp:96-118); only nz = 1 is restated. */
#define REXPO_EPSILON 0.001        /* src/robust_expo_smoothness.h:16 */
#define REXPO_MAXITER 300          /* src/robust_expo_methods.cpp:24 */
#define REXPO_SOR_W   1.9          /* :25 */
#define REXPO_XI      0.05         /* src/robust_expo_smoothness.cpp:17-19 */
#define REXPO_TAU     0.94
#define REXPO_BETA    0.001

/* src/operators.cpp:506-624 with boundary_condition = BOUNDARY_CONDITION_DIRICHLET: samples outside the image are 0 and
 * there is no size check (the padded line always has room) */
void orc_gaussian_dirichlet(double *I, int nx, int ny, double sigma)
{
    double *B;
    const int size = gauss_taps(sigma, &B);
    double *line = dalloc((size_t) (nx > ny ? nx : ny));
    for (int k = 0; k < ny; k++) {
        double *row = I + (size_t) k * nx;
        for (int x = 0; x < nx; x++) {
            double sum = B[0] * row[x];
            for (int j = 1; j < size; j++)
                sum += B[j] * ((x - j >= 0 ? row[x - j] : 0.0) + (x + j < nx ? row[x + j] : 0.0));
            line[x] = sum;
        }
        memcpy(row, line, (size_t) nx * sizeof(double));
    }
    for (int k = 0; k < nx; k++) {
        for (int y = 0; y < ny; y++) {
            double sum = B[0] * I[(size_t) y * nx + k];
            for (int j = 1; j < size; j++)
                sum += B[j] * ((y - j >= 0 ? I[(size_t) (y - j) * nx + k] : 0.0) + (y + j < ny ? I[(size_t) (y + j) * nx + k] : 0.0));
            line[y] = sum;
        }
        for (int y = 0; y < ny; y++) I[(size_t) y * nx + k] = line[y];
    }
    free(line);
    free(B);
}

static int rexpo_cmp(const void *a, const void *b)
{
    const double x = *(const double *) a, y = *(const double *) b;
    return (x > y) - (x < y);
}

/* src/robust_expo_smoothness.cpp:128-187 (nz = 1): expo = exp(-lambda |grad I1|) (+ BETA for method 2); method 3 picks lambda
 * per pixel, bounded by the value at the TAU quantile of the sorted gradients */
void orc_rexpo_exponential(const double *Ix, const double *Iy, int n, double alpha, double lambda, int method, double *expo)
{
    double *mg = dalloc((size_t) n);
    for (int i = 0; i < n; i++) mg[i] = sqrt(Ix[i] * Ix[i] + Iy[i] * Iy[i]);                       /* max_gradients, one channel */
    if (method == 1 || method == 2) {
        const double beta = method == 2 ? REXPO_BETA : 0;
        for (int i = 0; i < n; i++) expo[i] = exp(-lambda * mg[i]) + beta;
    } else if (method == 3) {
        double *lp = dalloc((size_t) n), *ord = dalloc((size_t) n);
        for (int i = 0; i < n; i++) lp[i] = (-log(REXPO_XI) + log(alpha)) / mg[i];
        memcpy(ord, mg, (size_t) n * sizeof(double));
        qsort(ord, (size_t) n, sizeof(double), rexpo_cmp);
        const double c = -log(REXPO_XI) + log(alpha);
        int pos_ref = (int) (REXPO_TAU * n);
        double lambda_omega;
        while ((pos_ref < n) && (c / 2 > ord[pos_ref - 1])) pos_ref++;
        if (pos_ref == n) lambda_omega = 0;
        else lambda_omega = (c / ord[pos_ref - 1]);
        for (int i = 0; i < n; i++) {
            double lambda_pi = lambda_omega;
            if (lambda_omega > lp[i]) lambda_pi = lp[i];
            expo[i] = exp(-lambda_pi * mg[i]);
        }
        free(lp); free(ord);
    }
    free(mg);
}

/* generic_tensor.cpp:97-168: existing terms in the order right, left, down, up */
static inline double rexpo_div_at(const double *f, const double *pr, const double *pl, const double *pd, const double *pu,
                                  int i, int j, int nx, int ny)
{
    const int k = i * nx + j;
    double acc = 0;
    int have = 0;
    if (j < nx - 1) { acc = pr[k] * (f[k + 1] - f[k]); have = 1; }
    if (j > 0)      { const double t = pl[k] * (f[k - 1] - f[k]);  acc = have ? acc + t : t; have = 1; }
    if (i < ny - 1) { const double t = pd[k] * (f[k + nx] - f[k]); acc = have ? acc + t : t; have = 1; }
    if (i > 0)      { const double t = pu[k] * (f[k - nx] - f[k]); acc = have ? acc + t : t; have = 1; }
    return acc;
}

/* methods.cpp:111-154; psi1..4 = right, left, down, up */
static inline double rexpo_sor_point(const double *Au, const double *Av, const double *Du, const double *Dv, const double *D,
                                     double *du, double *dv, double alpha, const double *psi1, const double *psi2,
                                     const double *psi3, const double *psi4, int i, int i0, int i1, int j, int nx, int j0, int j1)
{
    const double w = REXPO_SOR_W;
    const int k = i * nx + j;
    const double div_du = psi1[k] * du[k + j1] + psi2[k] * du[k - j0] + psi3[k] * du[k + i1] + psi4[k] * du[k - i0];
    const double div_dv = psi1[k] * dv[k + j1] + psi2[k] * dv[k - j0] + psi3[k] * dv[k + i1] + psi4[k] * dv[k - i0];
    const double duk = du[k], dvk = dv[k];
    du[k] = (1. - w) * du[k] + w * (Au[k] - D[k] * dv[k] + alpha * div_du) / Du[k];
    dv[k] = (1. - w) * dv[k] + w * (Av[k] - D[k] * du[k] + alpha * div_dv) / Dv[k];
    return (du[k] - duk) * (du[k] - duk) + (dv[k] - dvk) * (dv[k] - dvk);
}

/* methods.cpp:161-455, one channel, one thread (the reference's interior loop is a racy omp parallel for like Brox's) */
static void rexpo_single_scale(const double *I1, const double *I2, double *u, double *v, int nx, int ny, int method,
                               double alpha, double gamma, double lambda, double TOL, int inner_iter, int outer_iter,
                               int verbose, int *iters)
{
    const int size = nx * ny;
    const size_t n = (size_t) size;
    enum { NARR = 35 };
    double *a[NARR];
    for (int q = 0; q < NARR; q++) a[q] = dalloc(n);
    double *du = a[0], *dv = a[1], *ux = a[2], *uy = a[3], *vx = a[4], *vy = a[5];
    double *I1x = a[6], *I1y = a[7], *I2x = a[8], *I2y = a[9], *I2w = a[10], *I2wx = a[11], *I2wy = a[12];
    double *I2xx = a[13], *I2yy = a[14], *I2xy = a[15], *I2wxx = a[16], *I2wyy = a[17], *I2wxy = a[18];
    double *div_u = a[19], *div_v = a[20], *div_d = a[21];
    double *Au = a[22], *Av = a[23], *Du = a[24], *Dv = a[25], *D = a[26];
    double *psid = a[27], *psig = a[28], *psis = a[29], *psi1 = a[30], *psi2 = a[31], *psi3 = a[32], *psi4 = a[33], *expo = a[34];
    int solve = 0;

    orc_centered_gradient(I1, I1x, I1y, nx, ny);               /* :221-222 */
    orc_centered_gradient(I2, I2x, I2y, nx, ny);
    orc_dxx(I2, I2xx, nx, ny);                                 /* :225-227 */
    orc_dyy(I2, I2yy, nx, ny);
    orc_dxy(I2, I2xy, nx, ny);
    orc_rexpo_exponential(I1x, I1y, size, alpha, lambda, method, expo);     /* :231 */

    for (int no = 0; no < outer_iter; no++) {                  /* :234 */
        orc_bicubic_warp(I2,   u, v, I2w,   nx, ny, 1);        /* :236-241 (bicubic_interpolation_warp_color, nz = 1) */
        orc_bicubic_warp(I2x,  u, v, I2wx,  nx, ny, 1);
        orc_bicubic_warp(I2y,  u, v, I2wy,  nx, ny, 1);
        orc_bicubic_warp(I2xx, u, v, I2wxx, nx, ny, 1);
        orc_bicubic_warp(I2xy, u, v, I2wxy, nx, ny, 1);
        orc_bicubic_warp(I2yy, u, v, I2wyy, nx, ny, 1);
        orc_centered_gradient(u, ux, uy, nx, ny);              /* :244-245 */
        orc_centered_gradient(v, vx, vy, nx, ny);
        for (int i = 0; i < size; i++) {                       /* robust_expo_psi_smooth, smoothness.cpp:36-43 */
            const double gu = expo[i] * ux[i] * ux[i] + expo[i] * uy[i] * uy[i];
            const double gv = expo[i] * vx[i] * vx[i] + expo[i] * vy[i] * vy[i];
            const double normFlow = gu + gv;
            psis[i] = expo[i] / sqrt(normFlow + REXPO_EPSILON * REXPO_EPSILON);
        }
        for (int i = 0; i < ny; i++)                           /* robust_expo_psi_divergence */
            for (int j = 0; j < nx; j++) {
                const int k = i * nx + j;
                psi1[k] = (j < nx - 1) ? 0.5 * (psis[k + 1] + psis[k]) : 0;
                psi2[k] = (j > 0)      ? 0.5 * (psis[k - 1] + psis[k]) : 0;
                psi3[k] = (i < ny - 1) ? 0.5 * (psis[k + nx] + psis[k]) : 0;
                psi4[k] = (i > 0)      ? 0.5 * (psis[k - nx] + psis[k]) : 0;
            }
        for (int i = 0; i < ny; i++)                           /* robust_expo_divergence of u and of v, :257-258 */
            for (int j = 0; j < nx; j++) {
                div_u[i * nx + j] = rexpo_div_at(u, psi1, psi2, psi3, psi4, i, j, nx, ny);
                div_v[i * nx + j] = rexpo_div_at(v, psi1, psi2, psi3, psi4, i, j, nx, ny);
            }
        for (int i = 0; i < size; i++) {                       /* :261-267 */
            div_d[i] = alpha * (psi1[i] + psi2[i] + psi3[i] + psi4[i]);
            du[i] = dv[i] = 0;
        }
        for (int ni = 0; ni < inner_iter; ni++) {              /* :270 */
            for (int i = 0; i < size; i++) {                   /* psi_data :48-60, one channel */
                double dI2 = 0;
                const double dI = I2w[i] + I2wx[i] * du[i] + I2wy[i] * dv[i] - I1[i];
                dI2 += dI * dI;
                psid[i] = (1. / sqrt(dI2 + REXPO_EPSILON * REXPO_EPSILON));
            }
            for (int i = 0; i < size; i++) {                   /* psi_gradient :85-102 */
                double dI2 = 0;
                const double dIx = I2wx[i] + I2wxx[i] * du[i] + I2wxy[i] * dv[i] - I1x[i];
                const double dIy = I2wy[i] + I2wxy[i] * du[i] + I2wyy[i] * dv[i] - I1y[i];
                dI2 += dIx * dIx + dIy * dIy;
                psig[i] = (1. / sqrt(dI2 + REXPO_EPSILON * REXPO_EPSILON));
            }
            for (int i = 0; i < size; i++) {                   /* :279-322 */
                double BNu = 0, BNv = 0, BDu = 0, BDv = 0, GNu = 0, GNv = 0, GDu = 0, GDv = 0, DI_Gradient = 0, DI_Data = 0;
                const double dif = I2w[i] - I1[i];
                BNu += dif * I2wx[i];
                BNv += dif * I2wy[i];
                BDu += I2wx[i] * I2wx[i];
                BDv += I2wy[i] * I2wy[i];
                DI_Data += (I2wy[i] * I2wx[i]);
                const double dx = (I2wx[i] - I1x[i]);
                const double dy = (I2wy[i] - I1y[i]);
                GNu += (dx * I2wxx[i] + dy * I2wxy[i]);
                GNv += (dx * I2wxy[i] + dy * I2wyy[i]);
                GDu += (I2wxx[i] * I2wxx[i] + I2wxy[i] * I2wxy[i]);
                GDv += (I2wyy[i] * I2wyy[i] + I2wxy[i] * I2wxy[i]);
                DI_Gradient += (I2wxx[i] + I2wyy[i]) * I2wxy[i];
                const double g = gamma * psig[i];
                BNu = -psid[i] * BNu;
                BNv = -psid[i] * BNv;
                BDu = psid[i] * BDu;
                BDv = psid[i] * BDv;
                GNu = -g * GNu;
                GNv = -g * GNv;
                GDu = g * GDu;
                GDv = g * GDv;
                Au[i] = BNu + GNu + alpha * div_u[i];
                Av[i] = BNv + GNv + alpha * div_v[i];
                Du[i] = BDu + GDu + div_d[i];
                Dv[i] = BDv + GDv + div_d[i];
                D[i] = psid[i] * DI_Data + g * DI_Gradient;
            }
            double error = 1000;
            int nsor = 0;
            while (error > TOL && nsor < REXPO_MAXITER) {      /* :325-412, the sweep order of Brox */
                error = 0;
                nsor++;
                for (int i = 1; i < ny - 1; i++)
                    for (int j = 1; j < nx - 1; j++)
                        error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i, nx, nx, j, nx, 1, 1);
                for (int j = 1; j < nx - 1; j++) {
                    error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, 0, 0, nx, j, nx, 1, 1);
                    error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, ny - 1, nx, 0, j, nx, 1, 1);
                }
                for (int i = 1; i < ny - 1; i++) {
                    error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i, nx, nx, 0, nx, 0, 1);
                    error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, i, nx, nx, nx - 1, nx, 1, 0);
                }
                error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, 0, 0, nx, 0, nx, 0, 1);
                error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, 0, 0, nx, nx - 1, nx, 1, 0);
                error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, ny - 1, nx, 0, 0, nx, 0, 1);
                error += rexpo_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, ny - 1, nx, 0, nx - 1, nx, 1, 0);
                error = sqrt(error / size);                    /* :411 (size = pixels x channels) */
            }
            if (verbose) printf("Iterations: %d Error: %g\n", nsor, error);
            if (iters) iters[solve] = nsor;
            solve++;
        }
        for (int i = 0; i < size; i++) { u[i] += du[i]; v[i] += dv[i]; }   /* :419-422 */
    }
    for (int q = 0; q < NARR; q++) free(a[q]);
}

/* methods.cpp:463-567 with nzz = 1.  `iters` is laid out [scale][outer*inner].  Returns 1 where zoom_out's Gaussian throws. */
int orc_robust_expo(const double *I1, const double *I2, double *u, double *v, int nxx, int nyy, int method, double alpha,
                    double gamma, double lambda, int nscales, double nu, double TOL, int inner_iter, int outer_iter, int verbose,
                    int *iters)
{
    const int nzz = 1;
    pyramid P;
    P.nscales = nscales;
    P.nx = (int *) malloc(sizeof(int) * nscales);
    P.ny = (int *) malloc(sizeof(int) * nscales);
    P.A = (double **) calloc(nscales, sizeof(double *));
    P.B = (double **) calloc(nscales, sizeof(double *));
    P.u = (double **) calloc(nscales, sizeof(double *));
    P.v = (double **) calloc(nscales, sizeof(double *));
    P.nx[0] = nxx; P.ny[0] = nyy;
    P.A[0] = dalloc((size_t) nxx * nyy);
    P.B[0] = dalloc((size_t) nxx * nyy);
    P.u[0] = u; P.v[0] = v;
    int rc = 0;
    orc_image_normalization_2_color(I1, I2, P.A[0], P.B[0], nxx * nyy * nzz, nzz);     /* :494 */
    orc_gaussian_dirichlet(P.A[0], nxx, nyy, (double) nzz);                              /* :497-498: sigma = nzz, bc = (int) 0.8 */
    orc_gaussian_dirichlet(P.B[0], nxx, nyy, (double) nzz);
    for (int s = 1; s < nscales && !rc; s++) {
        orc_zoom_size(P.nx[s - 1], P.ny[s - 1], &P.nx[s], &P.ny[s], nu);
        const size_t n = (size_t) P.nx[s] * P.ny[s];
        P.A[s] = dalloc(n); P.B[s] = dalloc(n); P.u[s] = dalloc(n); P.v[s] = dalloc(n);
        rc |= orc_zoom_out(P.A[s - 1], P.A[s], P.nx[s - 1], P.ny[s - 1], nu);           /* zoom_out_color, nz = 1 */
        rc |= orc_zoom_out(P.B[s - 1], P.B[s], P.nx[s - 1], P.ny[s - 1], nu);
    }
    if (!rc) {
        const int c = nscales - 1;
        for (int i = 0; i < P.nx[c] * P.ny[c]; i++) P.u[c][i] = P.v[c][i] = 0.0;
    }
    const int alpha_adapted_for_nchannels = (int) (alpha * nzz);                         /* :529: an int */
    for (int s = nscales - 1; s >= 0 && !rc; s--) {
        if (verbose) printf("Scale: %d\n", s);
        rexpo_single_scale(P.A[s], P.B[s], P.u[s], P.v[s], P.nx[s], P.ny[s], method, alpha_adapted_for_nchannels, gamma, lambda,
                           TOL, inner_iter, outer_iter, verbose, iters ? iters + s * inner_iter * outer_iter : NULL);
        if (s) pyramid_upsample(&P, s, nu);
    }
    pyramid_free(&P);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Brox temporal: src/brox_optic_flow_temporal.cpp + src/brox_temporal_mask.cpp.
 * A sequence of `frames` images gives nz = frames - 1 flow fields coupled through a temporal smoothness term.
 * Arrays are frame-major: element k = f * nx * ny + i * nx + j. */

/* src/operators.cpp:413-499: per-frame centred gradient (same border rule as centered_gradient) + temporal
 * centred difference, one-sided x 0.5 in the first / last frame */
void orc_centered_gradient3(const double *in, double *dx, double *dy, double *dz, int nx, int ny, int nz)
{
    const int df = nx * ny;
    for (int f = 0; f < nz; f++) orc_centered_gradient(in + (size_t) f * df, dx + (size_t) f * df, dy + (size_t) f * df, nx, ny);
    if (nz > 1) {
        for (int f = 1; f < nz - 1; f++)
            for (int i = 0; i < df; i++) {
                const int k = f * df + i;
                dz[k] = 0.5 * (in[k + df] - in[k - df]);
            }
        for (int i = 0; i < df; i++) {
            int k = i;
            dz[k] = 0.5 * (in[k + df] - in[k]);
            k = (nz - 1) * df + i;
            dz[k] = 0.5 * (in[k] - in[k - df]);
        }
    } else {
        for (int i = 0; i < df; i++) dz[i] = 0;
    }
}

/* src/utils.cpp:251-276 + getminmax :509-525 */
void orc_image_normalization_1(const double *I, double *In, int size)
{
    double lo = I[0], hi = I[0];
    for (int i = 1; i < size; i++) { if (I[i] < lo) lo = I[i]; if (I[i] > hi) hi = I[i]; }
    const double den = hi - lo;
    if (den > 0) {
        #pragma omp parallel for
        for (int i = 0; i < size; i++) In[i] = 255.0 * (I[i] - lo) / den;
    } else {
        #pragma omp parallel for
        for (int i = 0; i < size; i++) In[i] = I[i];
    }
}

/* src/brox_temporal_mask.cpp:18-132: psi1..4 per frame as in the spatial method, psi5 / psi6 = half-sums with
 * the previous / following frame, 0 outside the sequence */
static void brox_t_psi_divergence(const double *psi, double *psi1, double *psi2, double *psi3, double *psi4,
                                  double *psi5, double *psi6, int nx, int ny, int nz)
{
    const int df = nx * ny;
    for (int f = 0; f < nz; f++)
        brox_psi_divergence(psi + (size_t) f * df, psi1 + (size_t) f * df, psi2 + (size_t) f * df, psi3 + (size_t) f * df,
                            psi4 + (size_t) f * df, nx, ny);
    if (nz > 1) {
        for (int f = 1; f < nz - 1; f++)
            for (int i = 0; i < df; i++) {
                const int k = f * df + i;
                psi5[k] = 0.5 * (psi[k - df] + psi[k]);
                psi6[k] = 0.5 * (psi[k + df] + psi[k]);
            }
        for (int i = 0; i < df; i++) {
            int k = i;
            psi5[k] = 0;
            psi6[k] = 0.5 * (psi[k + df] + psi[k]);
            k = (nz - 1) * df + i;
            psi5[k] = 0.5 * (psi[k - df] + psi[k]);
            psi6[k] = 0;
        }
    } else {
        for (int i = 0; i < df; i++) psi5[i] = psi6[i] = 0;
    }
}

/* src/brox_temporal_mask.cpp:140-239: the spatial divergence per frame, then `+=` the temporal terms (both as
 * ONE added expression in the interior frames) */
static void brox_t_divergence_u(const double *u, const double *v, const double *psi1, const double *psi2,
                                const double *psi3, const double *psi4, const double *psi5, const double *psi6,
                                double *div_u, double *div_v, int nx, int ny, int nz)
{
    const int df = nx * ny;
    for (int f = 0; f < nz; f++) {
        const size_t o = (size_t) f * df;
        brox_divergence_u(u + o, v + o, psi1 + o, psi2 + o, psi3 + o, psi4 + o, div_u + o, div_v + o, nx, ny);
    }
    if (nz > 1) {
        for (int f = 1; f < nz - 1; f++)
            for (int i = 0; i < df; i++) {
                const int k = f * df + i;
                div_u[k] += psi5[k] * (u[k - df] - u[k]) + psi6[k] * (u[k + df] - u[k]);
                div_v[k] += psi5[k] * (v[k - df] - v[k]) + psi6[k] * (v[k + df] - v[k]);
            }
        for (int i = 0; i < df; i++) {
            int k = i;
            div_u[k] += psi6[k] * (u[k + df] - u[k]);
            div_v[k] += psi6[k] * (v[k + df] - v[k]);
            k = (nz - 1) * df + i;
            div_u[k] += psi5[k] * (u[k - df] - u[k]);
            div_v[k] += psi5[k] * (v[k - df] - v[k]);
        }
    }
}

/* src/brox_optic_flow_temporal.cpp:120-170.  Offsets to the previous / following frame, row and column; 0 for a
 * missing neighbour (the tap lands on the pixel itself, with psi = 0 there). */
static inline double brox_t_sor_point(const double *Au, const double *Av, const double *Du, const double *Dv,
                                      const double *D, double *du, double *dv, double alpha, const double *psi1,
                                      const double *psi2, const double *psi3, const double *psi4, const double *psi5,
                                      const double *psi6, int f, int df0, int df1, int i, int ny, int dy0, int dy1,
                                      int j, int nx, int dx0, int dx1)
{
    const double w = BROX_SOR_W;
    const int k = f * ny * nx + i * nx + j;
    const double div_du = psi1[k] * du[k + dy1] + psi2[k] * du[k - dy0] + psi3[k] * du[k + dx1] + psi4[k] * du[k - dx0] +
                          psi5[k] * du[k - df0] + psi6[k] * du[k + df1];
    const double div_dv = psi1[k] * dv[k + dy1] + psi2[k] * dv[k - dy0] + psi3[k] * dv[k + dx1] + psi4[k] * dv[k - dx0] +
                          psi5[k] * dv[k - df0] + psi6[k] * dv[k + df1];
    const double duk = du[k], dvk = dv[k];
    du[k] = (1. - w) * du[k] + w * (Au[k] - D[k] * dv[k] + alpha * div_du) / Du[k];
    dv[k] = (1. - w) * dv[k] + w * (Av[k] - D[k] * du[k] + alpha * div_dv) / Dv[k];
    return (du[k] - duk) * (du[k] - duk) + (dv[k] - dvk) * (dv[k] - dvk);
}

/* src/brox_optic_flow_temporal.cpp:178-275: one frame of one sweep, the spatial method's visiting order */
static double brox_t_process_frame(const double *Au, const double *Av, const double *Du, const double *Dv,
                                   const double *D, double *du, double *dv, double alpha, const double *psi1,
                                   const double *psi2, const double *psi3, const double *psi4, const double *psi5,
                                   const double *psi6, int f, int nx, int ny, int df0, int df1)
{
#define BT(i, dy0, dy1, j, dx0, dx1) \
    brox_t_sor_point(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, psi5, psi6, f, df0, df1, i, ny, dy0, dy1, j, nx, dx0, dx1)
    double error = 0;
    #pragma omp parallel for reduction(+:error)
    for (int i = 1; i < ny - 1; i++)
        for (int j = 1; j < nx - 1; j++) error += BT(i, nx, nx, j, 1, 1);
    for (int j = 1; j < nx - 1; j++) {
        error += BT(0, 0, nx, j, 1, 1);
        error += BT(ny - 1, nx, 0, j, 1, 1);
    }
    for (int i = 1; i < ny - 1; i++) {
        error += BT(i, nx, nx, 0, 0, 1);
        error += BT(i, nx, nx, nx - 1, 1, 0);
    }
    error += BT(0, 0, nx, 0, 0, 1);
    error += BT(0, 0, nx, nx - 1, 1, 0);
    error += BT(ny - 1, nx, 0, 0, 0, 1);
    error += BT(ny - 1, nx, 0, nx - 1, 1, 0);
#undef BT
    return error;
}

/* src/brox_optic_flow_temporal.cpp:282-512 */
static void brox_t_single_scale(const double *I, double *u, double *v, int nx, int ny, int frames, double alpha,
                                double gamma, double TOL, int inner_iter, int outer_iter, int verbose, int *iters)
{
    const int nz = frames - 1, df = nx * ny;
    const int size = df * frames, size1 = df * nz;
    enum { NARR1 = 33 };
    double *a[NARR1];
    for (int q = 0; q < NARR1; q++) a[q] = dalloc((size_t) size1);
    double *du = a[0], *dv = a[1], *ux = a[2], *uy = a[3], *ut = a[4], *vx = a[5], *vy = a[6], *vt = a[7];
    double *Iw = a[8], *Iwx = a[9], *Iwy = a[10], *Ixx = a[11], *Iyy = a[12], *Ixy = a[13];
    double *Iwxx = a[14], *Iwyy = a[15], *Iwxy = a[16], *div_u = a[17], *div_v = a[18], *div_d = a[19];
    double *Au = a[20], *Av = a[21], *Du = a[22], *Dv = a[23], *D = a[24], *psid = a[25], *psig = a[26], *psis = a[27];
    double *psi1 = a[28], *psi2 = a[29], *psi3 = a[30], *psi4 = a[31], *psi5 = a[32];
    double *psi6 = dalloc((size_t) size1), *Ix = dalloc((size_t) size), *Iy = dalloc((size_t) size);
    int solve = 0;

    for (int f = 0; f < frames; f++) orc_centered_gradient(I + (size_t) f * df, Ix + (size_t) f * df, Iy + (size_t) f * df, nx, ny);   /* :346-348 */
    for (int f = 0; f < nz; f++) {                                                    /* :351-355 */
        orc_dxx(I + (size_t) df * (f + 1), Ixx + (size_t) f * df, nx, ny);
        orc_dyy(I + (size_t) df * (f + 1), Iyy + (size_t) f * df, nx, ny);
        orc_dxy(I + (size_t) df * (f + 1), Ixy + (size_t) f * df, nx, ny);
    }
    for (int no = 0; no < outer_iter; no++) {                                         /* :358 */
        for (int f = 0; f < nz; f++) {                                                /* :360-367 */
            const size_t o = (size_t) f * df, o1 = (size_t) df * (f + 1);
            orc_bicubic_warp(I + o1,  u + o, v + o, Iw + o,   nx, ny, 1);
            orc_bicubic_warp(Ix + o1, u + o, v + o, Iwx + o,  nx, ny, 1);
            orc_bicubic_warp(Iy + o1, u + o, v + o, Iwy + o,  nx, ny, 1);
            orc_bicubic_warp(Ixx + o, u + o, v + o, Iwxx + o, nx, ny, 1);
            orc_bicubic_warp(Ixy + o, u + o, v + o, Iwxy + o, nx, ny, 1);
            orc_bicubic_warp(Iyy + o, u + o, v + o, Iwyy + o, nx, ny, 1);
        }
        orc_centered_gradient3(u, ux, uy, ut, nx, ny, nz);                           /* :370-371 */
        orc_centered_gradient3(v, vx, vy, vt, nx, ny, nz);
        #pragma omp parallel for
        for (int i = 0; i < size1; i++) {                                             /* psi_smooth :94-118 */
            const double gu = ux[i] * ux[i] + uy[i] * uy[i] + ut[i] * ut[i];
            const double gv = vx[i] * vx[i] + vy[i] * vy[i] + vt[i] * vt[i];
            const double d2 = gu + gv;
            psis[i] = 1. / sqrt(d2 + BROX_EPSILON * BROX_EPSILON);
        }
        brox_t_psi_divergence(psis, psi1, psi2, psi3, psi4, psi5, psi6, nx, ny, nz);                 /* :377 */
        brox_t_divergence_u(u, v, psi1, psi2, psi3, psi4, psi5, psi6, div_u, div_v, nx, ny, nz);     /* :380 */
        #pragma omp parallel for
        for (int i = 0; i < size1; i++) {                                             /* :383-390 */
            div_d[i] = alpha * (psi1[i] + psi2[i] + psi3[i] + psi4[i] + psi5[i] + psi6[i]);
            du[i] = dv[i] = 0;
        }
        for (int ni = 0; ni < inner_iter; ni++) {                                     /* :394 */
            #pragma omp parallel for
            for (int i = 0; i < size1; i++) {                                         /* psi_data :36-55 */
                const double dI = Iw[i] - I[i] + Iwx[i] * du[i] + Iwy[i] * dv[i];
                const double dI2 = dI * dI;
                psid[i] = 1. / sqrt(dI2 + BROX_EPSILON * BROX_EPSILON);
            }
            #pragma omp parallel for
            for (int i = 0; i < size1; i++) {                                         /* psi_gradient :63-86 */
                const double dIx = Iwx[i] - Ix[i] + Iwxx[i] * du[i] + Iwxy[i] * dv[i];
                const double dIy = Iwy[i] - Iy[i] + Iwxy[i] * du[i] + Iwyy[i] * dv[i];
                const double dI2 = dIx * dIx + dIy * dIy;
                psig[i] = 1. / sqrt(dI2 + BROX_EPSILON * BROX_EPSILON);
            }
            for (int i = 0; i < size1; i++) {                                         /* :400-427 */
                const double p = psid[i];
                const double g = gamma * psig[i];
                const double dif = Iw[i] - I[i];
                const double BNu = -p * dif * Iwx[i];
                const double BNv = -p * dif * Iwy[i];
                const double BDu = p * Iwx[i] * Iwx[i];
                const double BDv = p * Iwy[i] * Iwy[i];
                const double dx = (Iwx[i] - Ix[i]);
                const double dy = (Iwy[i] - Iy[i]);
                const double GNu = -g * (dx * Iwxx[i] + dy * Iwxy[i]);
                const double GNv = -g * (dx * Iwxy[i] + dy * Iwyy[i]);
                const double GDu = g * (Iwxx[i] * Iwxx[i] + Iwxy[i] * Iwxy[i]);
                const double GDv = g * (Iwyy[i] * Iwyy[i] + Iwxy[i] * Iwxy[i]);
                const double DI = (Iwxx[i] + Iwyy[i]) * Iwxy[i];
                const double Duv = p * Iwy[i] * Iwx[i] + g * DI;
                Au[i] = BNu + GNu + alpha * div_u[i];
                Av[i] = BNv + GNv + alpha * div_v[i];
                Du[i] = BDu + GDu + div_d[i];
                Dv[i] = BDv + GDv + div_d[i];
                D[i] = Duv;
            }
            double error = 1000;
            int nsor = 0;
            while (error > TOL && nsor < BROX_MAXITER) {                              /* :433-461 */
                error = 0;
                nsor++;
                for (int f = 1; f < nz - 1; f++)                                      /* interior frames first */
                    error += brox_t_process_frame(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, psi5, psi6, f,
                                                  nx, ny, df, df);
                error += brox_t_process_frame(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, psi5, psi6, 0, nx,
                                              ny, 0, df);
                error += brox_t_process_frame(Au, Av, Du, Dv, D, du, dv, alpha, psi1, psi2, psi3, psi4, psi5, psi6,
                                              nz - 1, nx, ny, df, 0);
                error = sqrt(error / size1);
            }
            if (verbose) printf("Iterations: %d\n", nsor);
            if (iters) iters[solve] = nsor;
            solve++;
        }
        for (int i = 0; i < size1; i++) { u[i] += du[i]; v[i] += dv[i]; }            /* :469-472 */
    }
    for (int q = 0; q < NARR1; q++) free(a[q]);
    free(psi6); free(Ix); free(Iy);
}

/* src/brox_optic_flow_temporal.cpp:520-627.  u, v: (frames - 1) * nx * ny; `iters` is laid out
 * [scale][outer*inner].  Returns 1 where the reference throws ("sigma too large"), 2 for frames <= 2 (the
 * reference prints a message and returns without touching u, v). */
int orc_brox_temporal(const double *I, double *u, double *v, int nxx, int nyy, int frames, double alpha, double gamma,
                      int nscales, double nu, double TOL, int inner_iter, int outer_iter, int verbose, int *iters)
{
    if (frames <= 2) return 2;
    int *nx = (int *) malloc(sizeof(int) * nscales), *ny = (int *) malloc(sizeof(int) * nscales);
    double **Is = (double **) calloc(nscales, sizeof(double *));
    double **us = (double **) calloc(nscales, sizeof(double *)), **vs = (double **) calloc(nscales, sizeof(double *));
    int rc = 0;
    nx[0] = nxx; ny[0] = nyy;
    Is[0] = dalloc((size_t) nxx * nyy * frames);
    orc_image_normalization_1(I, Is[0], nxx * nyy * frames);                          /* :548 */
    for (int f = 0; f < frames; f++) rc |= orc_gaussian(Is[0] + (size_t) f * nxx * nyy, nxx, nyy, BROX_SIGMA);   /* :551-553 */
    us[0] = u; vs[0] = v;
    for (int s = 1; s < nscales && !rc; s++) {                                        /* :561-575 */
        orc_zoom_size(nx[s - 1], ny[s - 1], &nx[s], &ny[s], nu);
        const size_t n = (size_t) nx[s] * ny[s];
        Is[s] = dalloc(n * frames);
        us[s] = dalloc(n * (frames - 1));
        vs[s] = dalloc(n * (frames - 1));
        for (int f = 0; f < frames; f++)
            rc |= orc_zoom_out(Is[s - 1] + (size_t) f * nx[s - 1] * ny[s - 1], Is[s] + f * n, nx[s - 1], ny[s - 1], nu);
    }
    if (!rc) {
        const int c = nscales - 1;
        for (int i = 0; i < nx[c] * ny[c] * (frames - 1); i++) us[c][i] = vs[c][i] = 0.0;           /* :578-580 */
        for (int s = nscales - 1; s >= 0; s--) {                                      /* :587-614 */
            if (verbose) printf("Scale: %d\n", s);
            brox_t_single_scale(Is[s], us[s], vs[s], nx[s], ny[s], frames, alpha, gamma, TOL, inner_iter, outer_iter,
                                verbose, iters ? iters + s * inner_iter * outer_iter : NULL);
            if (s) {
                const size_t n = (size_t) nx[s] * ny[s], n1 = (size_t) nx[s - 1] * ny[s - 1];
                for (int f = 0; f < frames - 1; f++) {
                    orc_zoom_in(us[s] + f * n, us[s - 1] + f * n1, nx[s], ny[s], nx[s - 1], ny[s - 1]);
                    orc_zoom_in(vs[s] + f * n, vs[s - 1] + f * n1, nx[s], ny[s], nx[s - 1], ny[s - 1]);
                }
                for (size_t i = 0; i < n1 * (frames - 1); i++) {
                    us[s - 1][i] *= 1.0 / nu;
                    vs[s - 1][i] *= 1.0 / nu;
                }
            }
        }
    }
    for (int s = 0; s < nscales; s++) {
        free(Is[s]);
        if (s) { free(us[s]); free(vs[s]); }
    }
    free(Is); free(us); free(vs); free(nx); free(ny);
    return rc;
}

/* src/bicubic_interpolation.cpp:253-344: channel k of an image with nz interleaved channels; taps exactly as in
 * orc_bicubic_at (same truncation, sign and clamping quirks) */
double orc_bicubic_at_color(const double *in, double uu, double vv, int nx, int ny, int nz, int k, int border_out)
{
    const int sx = (uu < 0) ? -1 : 1, sy = (vv < 0) ? -1 : 1;
    int out = 0;
    const int x = clamp_flag((int) uu, nx, &out), y = clamp_flag((int) vv, ny, &out);
    const int mx = clamp_flag((int) uu - sx, nx, &out), my = clamp_flag((int) vv - sx, ny, &out);
    const int dx = clamp_flag((int) uu + sx, nx, &out), dy = clamp_flag((int) vv + sy, ny, &out);
    const int ddx = clamp_flag((int) uu + 2 * sx, nx, &out), ddy = clamp_flag((int) vv + 2 * sy, ny, &out);
    if (out && border_out) return 0.0;
    const int cols[4] = {mx, x, dx, ddx}, rows[4] = {my, y, dy, ddy};
    double c[4];
    for (int q = 0; q < 4; q++)
        c[q] = cubic_cell(in[((size_t) cols[q] + (size_t) nx * rows[0]) * nz + k], in[((size_t) cols[q] + (size_t) nx * rows[1]) * nz + k],
                          in[((size_t) cols[q] + (size_t) nx * rows[2]) * nz + k], in[((size_t) cols[q] + (size_t) nx * rows[3]) * nz + k],
                          vv - y);
    return cubic_cell(c[0], c[1], c[2], c[3], uu - x);
}

/* src/utils.cpp:509-525 */
void orc_getminmax(double *mn, double *mx, const double *x, int n)
{
    *mn = *mx = x[0];
    for (int i = 1; i < n; i++) {
        if (x[i] < *mn) *mn = x[i];
        if (x[i] > *mx) *mx = x[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Classic Horn-Schunck, src/horn_schunck_classic.cpp: n Jacobi iterations from a zero flow.  p(x, i, j) is the
 * sample at column i, row j with clamped (Neumann) indices (:22-43). */
static inline double hsc_p(const double *x, int w, int h, int i, int j)
{
    if (i < 0) i = 0;
    if (j < 0) j = 0;
    if (i >= w) i = w - 1;
    if (j >= h) j = h - 1;
    return x[j * w + i];
}

void orc_hs_classic(double *u, double *v, const double *a, const double *b, int w, int h, int n, double alpha)
{
    const size_t sz = (size_t) w * h;
    double *Ex = dalloc(sz), *Ey = dalloc(sz), *Et = dalloc(sz), *ubar = dalloc(sz), *vbar = dalloc(sz);
    for (int j = 0; j < h; j++)                                                       /* :46-73 */
        for (int i = 0; i < w; i++) {
            Ey[j * w + i] = (1.0 / 4) * (hsc_p(a, w, h, i, j + 1) - hsc_p(a, w, h, i, j) + hsc_p(a, w, h, i + 1, j + 1) -
                                         hsc_p(a, w, h, i + 1, j) + hsc_p(b, w, h, i, j + 1) - hsc_p(b, w, h, i, j) +
                                         hsc_p(b, w, h, i + 1, j + 1) - hsc_p(b, w, h, i + 1, j));
            Ex[j * w + i] = (1.0 / 4) * (hsc_p(a, w, h, i + 1, j) - hsc_p(a, w, h, i, j) + hsc_p(a, w, h, i + 1, j + 1) -
                                         hsc_p(a, w, h, i, j + 1) + hsc_p(b, w, h, i + 1, j) - hsc_p(b, w, h, i, j) +
                                         hsc_p(b, w, h, i + 1, j + 1) - hsc_p(b, w, h, i, j + 1));
            Et[j * w + i] = (1.0 / 4) * (hsc_p(b, w, h, i, j) - hsc_p(a, w, h, i, j) + hsc_p(b, w, h, i + 1, j) -
                                         hsc_p(a, w, h, i + 1, j) + hsc_p(b, w, h, i, j + 1) - hsc_p(a, w, h, i, j + 1) +
                                         hsc_p(b, w, h, i + 1, j + 1) - hsc_p(a, w, h, i + 1, j + 1));
        }
    for (size_t i = 0; i < sz; i++) u[i] = v[i] = 0;                                  /* :139-141 */
    for (int it = 0; it < n; it++) {                                                  /* hs_iteration :97-122 */
        for (int pass = 0; pass < 2; pass++) {                                        /* compute_bar :76-94 */
            const double *x = pass ? v : u;
            double *bar = pass ? vbar : ubar;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++)
                    bar[j * w + i] = (1.0 / 6) * (hsc_p(x, w, h, i - 1, j) + hsc_p(x, w, h, i + 1, j) + hsc_p(x, w, h, i, j - 1) +
                                                  hsc_p(x, w, h, i, j + 1)) +
                                     (1.0 / 12) * (hsc_p(x, w, h, i - 1, j - 1) + hsc_p(x, w, h, i + 1, j - 1) +
                                                   hsc_p(x, w, h, i - 1, j + 1) + hsc_p(x, w, h, i + 1, j + 1));
        }
        for (size_t i = 0; i < sz; i++) {
            double t = Ex[i] * ubar[i] + Ey[i] * vbar[i] + Et[i];
            t /= alpha * alpha + Ex[i] * Ex[i] + Ey[i] * Ey[i];
            u[i] = ubar[i] - Ex[i] * t;
            v[i] = vbar[i] - Ey[i] * t;
        }
    }
    free(Ex); free(Ey); free(Et); free(ubar); free(vbar);
}

/* ------------------------------------------------------------------------------------------ */
/* colour operators.  src/bicubic_interpolation.cpp:381-405 -- channel by channel, every sample through
 * bicubic_interpolation_at_color; (double)(j + u[p]) is j + u[p] evaluated in double. */
void orc_bicubic_warp_color(const double *in, const double *u, const double *v, double *out, int nx, int ny, int nz,
                            int border_out)
{
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < ny; i++)
            for (int j = 0; j < nx; j++) {
                const int p = i * nx + j;
                out[(size_t) p * nz + k] = orc_bicubic_at_color(in, j + u[p], i + v[p], nx, ny, nz, k, border_out);
            }
}

/* src/utils.cpp:333-404.  `size` is the element count of the interleaved arrays; channel c starts from element c and the
 * scan visits i = nz, 2 nz, ... < size (so the last pixel is read at i + c, as in the reference). */
void orc_image_normalization_2_color(const double *I1, const double *I2, double *I1n, double *I2n, int size, int nz)
{
    for (int c = 0; c < nz; c++) {
        double max = I1[c] > I2[c] ? I1[c] : I2[c];
        double min = I1[c] < I2[c] ? I1[c] : I2[c];
        for (int i = nz; i < size; i += nz) {
            const int r = i + c;
            if (I1[r] > max) max = I1[r];
            if (I1[r] < min) min = I1[r];
            if (I2[r] > max) max = I2[r];
            if (I2[r] < min) min = I2[r];
        }
        const double den = max - min;
        for (int i = 0; i < size; i += nz) {
            const int r = i + c;
            if (den > 0) {
                I1n[r] = 255.0 * (I1[r] - min) / den;
                I2n[r] = 255.0 * (I2[r] - min) / den;
            } else {
                I1n[r] = I1[r];
                I2n[r] = I2[r];
            }
        }
    }
}

/* src/utils.cpp:412-450: joint min/max of three images, in place, NO den > 0 test (a constant triple divides by zero) */
void orc_image_normalization_3(double *I0, double *I1, double *I2, int size)
{
    double min0, max0, min1, max1, min2, max2;
    orc_getminmax(&min0, &max0, I0, size);
    orc_getminmax(&min1, &max1, I1, size);
    orc_getminmax(&min2, &max2, I2, size);
    double max = max0, min = min0;
    if (max1 > max) max = max1;
    if (min1 < min) min = min1;
    if (max2 > max) max = max2;
    if (min2 < min) min = min2;
    const double den = max - min;
    for (int i = 0; i < size; i++) {
        I0[i] = 255.0 * (I0[i] - min) / den;
        I1[i] = 255.0 * (I1[i] - min) / den;
        I2[i] = 255.0 * (I2[i] - min) / den;
    }
}

/* src/utils.cpp:452-501 */
void orc_image_normalization_4(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *I_1n,
                               double *I0n, double *I1n, double *filtI0n, int size)
{
    double min_1, max_1, min0, max0, min1, max1, minf, maxf;
    orc_getminmax(&min_1, &max_1, I_1, size);
    orc_getminmax(&min0, &max0, I0, size);
    orc_getminmax(&min1, &max1, I1, size);
    orc_getminmax(&minf, &maxf, filtI0, size);
    double max = (max_1 > max0) ? max_1 : max0;
    max = (max > max1) ? max : max1;
    max = (max > maxf) ? max : maxf;
    double min = (min_1 < min0) ? min_1 : min0;
    min = (min < min1) ? min : min1;
    min = (min < minf) ? min : minf;
    const double den = max - min;
    for (int i = 0; i < size; i++) {
        if (den > 0) {
            I_1n[i] = 255.0 * (I_1[i] - min) / den;
            I0n[i] = 255.0 * (I0[i] - min) / den;
            I1n[i] = 255.0 * (I1[i] - min) / den;
            filtI0n[i] = 255.0 * (filtI0[i] - min) / den;
        } else {
            I_1n[i] = I_1[i];
            I0n[i] = I0[i];
            I1n[i] = I1[i];
            filtI0n[i] = filtI0[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* the deterministic building blocks of TV-L1 with occlusions.
 * src/utils.cpp:150-213 me_median_filtering: wsize x wsize window, indices mirrored WITH repetition of the border
 * sample (-1 -> 0, n -> n - 1), element [count / 2] of the sorted window (the sort order among equal values cannot
 * matter), in place through a copy. */
static int median_cmp(const void *a, const void *b)
{
    const double x = *(const double *) a, y = *(const double *) b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

void orc_median_filtering(double *in, int nx, int ny, int wsize)
{
    const int border = wsize >> 1;
    double *win = dalloc((size_t) wsize * wsize), *out = dalloc((size_t) nx * ny);
    for (int x = 0; x < nx; x++)
        for (int y = 0; y < ny; y++) {
            int n = 0;
            for (int yy = y - border; yy <= y + border; yy++)
                for (int xx = x - border; xx <= x + border; xx++) {
                    int x0 = xx, y0 = yy;
                    if (x0 < 0) x0 = -x0 - 1;
                    if (x0 >= nx) x0 = 2 * nx - x0 - 1;
                    if (y0 < 0) y0 = -y0 - 1;
                    if (y0 >= ny) y0 = 2 * ny - y0 - 1;
                    win[n++] = in[y0 * nx + x0];
                }
            qsort(win, (size_t) n, sizeof(double), median_cmp);
            out[y * nx + x] = win[n / 2];
        }
    memcpy(in, out, sizeof(double) * (size_t) nx * ny);
    free(win);
    free(out);
}

#define OCC_IS_ZERO 1E-10      /* src/tvl1occflow_constants.h:31 */
#define OCC_THR_CHI 0.75       /* src/tvl1occflow_constants.h:32 */

/* src/tvl1occflow_solvers.cpp:56-147 Solver_wrt_v: pointwise thresholding of the forward and the backward data term */
void orc_occ_solver_v(const double *u1, const double *u2, double *v1, double *v2, const double *chi, const double *I1wx,
                      const double *I1wy, const double *I_1wx, const double *I_1wy, const double *rho1_c,
                      const double *rho3_c, double *Vfwd_1, double *Vfwd_2, double *Vbck_1, double *Vbck_2,
                      const double *grad1, const double *grad3, double alpha, double theta, double lambda, int nx, int ny)
{
    const int size = nx * ny;
    const double l_t = lambda * theta;
    const double _1pat = 1. + alpha * theta;
    const double at_d_1pat = alpha * theta / _1pat;
    const double lt_d_1pat = 2. * lambda * theta / _1pat;
    for (int i = 0; i < size; i++) {
        double d1 = 0, d2 = 0;
        const double rho1 = rho1_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
        if (rho1 < -l_t * grad1[i]) {
            d1 = l_t * I1wx[i];
            d2 = l_t * I1wy[i];
        } else if (rho1 > l_t * grad1[i]) {
            d1 = -l_t * I1wx[i];
            d2 = -l_t * I1wy[i];
        } else if (grad1[i] < OCC_IS_ZERO) {
            d1 = d2 = 0;
        } else {
            d1 = -rho1 * I1wx[i] / grad1[i];
            d2 = -rho1 * I1wy[i] / grad1[i];
        }
        Vfwd_1[i] = u1[i] + d1;
        Vfwd_2[i] = u2[i] + d2;

        const double rho3 = rho3_c[i] - (I_1wx[i] * u1[i] + I_1wy[i] * u2[i]);
        const double A = rho3 + at_d_1pat * (I_1wx[i] * u1[i] + I_1wy[i] * u2[i]);
        if (A < -lt_d_1pat * grad3[i]) {
            d1 = -lt_d_1pat * I_1wx[i];
            d2 = -lt_d_1pat * I_1wy[i];
            Vbck_1[i] = (u1[i] / _1pat) + d1;
            Vbck_2[i] = (u2[i] / _1pat) + d2;
        } else if (A > lt_d_1pat * grad3[i]) {
            d1 = lt_d_1pat * I_1wx[i];
            d2 = lt_d_1pat * I_1wy[i];
            Vbck_1[i] = (u1[i] / _1pat) + d1;
            Vbck_2[i] = (u2[i] / _1pat) + d2;
        } else {
            if (grad3[i] < OCC_IS_ZERO) {
                d1 = d2 = 0;
            } else {
                d1 = rho3 * I_1wx[i] / grad3[i];
                d2 = rho3 * I_1wy[i] / grad3[i];
            }
            Vbck_1[i] = u1[i] + d1;
            Vbck_2[i] = u2[i] + d2;
        }
        if (chi[i] < OCC_THR_CHI) {
            v1[i] = Vfwd_1[i];
            v2[i] = Vfwd_2[i];
        } else {
            v1[i] = Vbck_1[i];
            v2[i] = Vbck_2[i];
        }
    }
}

/* src/tvl1occflow_solvers.cpp:218-337 Solver_wrt_chi with the dual variable (eta1, eta2) as EXPLICIT state.  The
 * reference keeps eta in function-local statics that it never initialises (its own "#warning eta1 and eta2 are used
 * uninitialized"); n_iter is its MAX_ITERATIONS_CHI (100).  Per iteration: eta += tau_eta g grad(chi), projection onto
 * the unit ball (:33-53), chi += tau_chi (div(g eta) - F - G - beta div u), clamped to [0, 1]. */
void orc_occ_solver_chi(const double *u1, const double *u2, double *chi, const double *I1wx, const double *I1wy,
                        const double *I_1wx, const double *I_1wy, const double *rho1_c, const double *rho3_c,
                        const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1, const double *Vbck_2,
                        const double *g, double lambda, double theta, double alpha, double beta, double tau_chi,
                        double tau_eta, int nx, int ny, double *eta1, double *eta2, int n_iter)
{
    const int size = nx * ny;
    double *chix = dalloc((size_t) size), *chiy = dalloc((size_t) size), *geta1 = dalloc((size_t) size);
    double *geta2 = dalloc((size_t) size), *div_eta = dalloc((size_t) size), *div_u = dalloc((size_t) size);
    for (int n_chi = 0; n_chi < n_iter; n_chi++) {
        orc_forward_gradient(chi, chix, chiy, nx, ny);
        for (int i = 0; i < size; i++) {
            eta1[i] = eta1[i] + tau_eta * g[i] * chix[i];
            eta2[i] = eta2[i] + tau_eta * g[i] * chiy[i];
        }
        for (int j = 0; j < size; j++) {                                           /* project, :33-53 */
            const double norm2 = eta1[j] * eta1[j] + eta2[j] * eta2[j];
            if (norm2 < OCC_IS_ZERO) {
                eta1[j] = 0.0;
                eta2[j] = 0.0;
            } else {
                const double norm = sqrt(norm2);
                eta1[j] = eta1[j] / norm;
                eta2[j] = eta2[j] / norm;
            }
        }
        for (int i = 0; i < size; i++) {
            geta1[i] = g[i] * eta1[i];
            geta2[i] = g[i] * eta2[i];
        }
        orc_divergence(geta1, geta2, div_eta, nx, ny);
        orc_divergence(u1, u2, div_u, nx, ny);
        for (int i = 0; i < size; i++) {
            const double rho1 = rho1_c[i] + (I1wx[i] * Vfwd_1[i] + I1wy[i] * Vfwd_2[i]);
            const double abs_rho1 = (rho1 < 0.) ? -rho1 : rho1;
            const double rho3 = rho3_c[i] - (I_1wx[i] * Vbck_1[i] + I_1wy[i] * Vbck_2[i]);
            const double abs_rho3 = (rho3 < 0.) ? -rho3 : rho3;
            double F, G;
            if (chi[i] < 0.5) {
                F = -lambda * abs_rho1;
                G = -(0.5 / theta) * ((Vfwd_1[i] - u1[i]) * (Vfwd_1[i] - u1[i]) + (Vfwd_2[i] - u2[i]) * (Vfwd_2[i] - u2[i]));
            } else {
                F = lambda * abs_rho3;
                G = (0.5 / theta) * ((Vbck_1[i] - u1[i]) * (Vbck_1[i] - u1[i]) + (Vbck_2[i] - u2[i]) * (Vbck_2[i] - u2[i]))
                    + alpha * theta * (Vbck_1[i] * Vbck_1[i] + Vbck_2[i] * Vbck_2[i]);
            }
            chi[i] = chi[i] + tau_chi * (div_eta[i] - F - G - beta * div_u[i]);
            if (chi[i] > 1.) chi[i] = 1.;
            else if (chi[i] < 0.) chi[i] = 0.;
        }
    }
    free(chix); free(chiy); free(geta1); free(geta2); free(div_eta); free(div_u);
}

/* ------------------------------------------------------------------------------------------ */
/* src/tvl1occflow_tv_rof_box.cpp:22-645 Scalar_ROF_BoxCellCentered: nIter times { alfa = |grad u| / (lambda g) per cell;
 * one in-place box relaxation sweep over the cells in lexicographic order; u = lambda f + lambda div P }.
 * Staggered (2 ny + 1) x (2 nx + 1) grid as in the reference: cell (ci, cj) has its centre at (2 ci + 1, 2 cj + 1), its
 * dual values on the four edges around it; P of the south / east edge of a cell is the in/out state initialP1 / initialP2,
 * the edges on the image border stay as initialised.  Nine cell kinds (corners, sides, inner), each with the
 * reference's own closed-form solve of its 2x2 / 3x3 / 4x4 system -- the expressions below keep its association order
 * (note: only the north side writes the free terms with the F term first).  nx, ny >= 2. */
#define SG(i, j) ((size_t) (i) * nxs + (j))
void orc_rof_box(double *u, const double *f, double *P1, double *P2, const double *g, double lambda, double omega, int nx,
                 int ny, int n_iter)
{
    const int nxs = 2 * nx + 1, nys = 2 * ny + 1;
    const size_t ns = (size_t) nxs * nys;
    double *P = dalloc(ns), *F = dalloc(ns), *AL = dalloc(ns);
    double *ux = dalloc((size_t) nx * ny), *uy = dalloc((size_t) nx * ny);
    for (size_t k = 0; k < ns; k++) P[k] = F[k] = AL[k] = 0.0;
    for (int ci = 0, q = 0; ci < ny; ci++)                                               /* :123-135 */
        for (int cj = 0; cj < nx; cj++, q++) {
            const int i = 2 * ci + 1, j = 2 * cj + 1;
            F[SG(i, j)] = f[q];
            P[SG(i + 1, j)] = P1[q];
            P[SG(i, j + 1)] = P2[q];
        }
    for (int i = 2; i <= nys - 3; i += 2)                                                /* :142-150 */
        for (int j = 1; j <= nxs - 2; j += 2) F[SG(i, j)] = F[SG(i + 1, j)] - F[SG(i - 1, j)];
    for (int i = 1; i <= nys - 2; i += 2)                                                /* :156-164 */
        for (int j = 2; j <= nxs - 3; j += 2) F[SG(i, j)] = F[SG(i, j + 1)] - F[SG(i, j - 1)];
    const double w = omega;
    for (int iter = 1; iter <= n_iter; iter++) {
        orc_forward_gradient(u, ux, uy, nx, ny);                                         /* :173 */
        for (int ci = 0, q = 0; ci < ny; ci++)
            for (int cj = 0; cj < nx; cj++, q++) {
                const int i = 2 * ci + 1, j = 2 * cj + 1;
                AL[SG(i + 1, j)] = sqrt(ux[q] * ux[q] + uy[q] * uy[q]) / (lambda * g[q]);    /* file-local hypot, :15-20,180 */
                AL[SG(i, j + 1)] = AL[SG(i + 1, j)];
            }
        for (int ci = 0; ci < ny; ci++)
            for (int cj = 0; cj < nx; cj++) {
                const int i = 2 * ci + 1, j = 2 * cj + 1;
                const int top = ci == 0, bot = ci == ny - 1, lef = cj == 0, rig = cj == nx - 1;
                const size_t jm1 = SG(i, j - 1), jp1 = SG(i, j + 1), im1 = SG(i - 1, j), ip1 = SG(i + 1, j);
                /* free terms; an operand that does not exist for this cell kind is never used below */
                double W = 0, N = 0, S = 0, E = 0;
                const int nside = top && !lef && !rig;
                if (!lef) W = nside ? -F[jm1] - P[SG(i, j - 3)] + P[SG(i + 1, j - 2)] - P[SG(i - 1, j - 2)]
                                    : -P[SG(i, j - 3)] + P[SG(i + 1, j - 2)] - P[SG(i - 1, j - 2)] - F[jm1];
                if (!top) N = -P[SG(i - 3, j)] + P[SG(i - 2, j + 1)] - P[SG(i - 2, j - 1)] - F[im1];
                if (!bot) S = nside ? -F[ip1] - P[SG(i + 3, j)] - P[SG(i + 2, j + 1)] + P[SG(i + 2, j - 1)]
                                    : -P[SG(i + 3, j)] - P[SG(i + 2, j + 1)] + P[SG(i + 2, j - 1)] - F[ip1];
                if (!rig) E = nside ? -F[jp1] - P[SG(i, j + 3)] - P[SG(i + 1, j + 2)] + P[SG(i - 1, j + 2)]
                                    : -P[SG(i, j + 3)] - P[SG(i + 1, j + 2)] + P[SG(i - 1, j + 2)] - F[jp1];
                const double b0 = lef ? 0.0 : -2 - AL[jm1], b1 = top ? 0.0 : -2 - AL[im1];
                const double b2 = bot ? 0.0 : -2 - AL[ip1], b3 = rig ? 0.0 : -2 - AL[jp1];
                double den;
                if (top && lef) {                                                        /* :189-218 */
                    den = b2 * b3 - 1;
                    const double ns_ = (1 - w) * P[ip1] + w * (S * b3 + E) / den;
                    const double ne_ = (1 - w) * P[jp1] + w * (E * b2 + S) / den;
                    P[ip1] = ns_; P[jp1] = ne_;
                } else if (top && rig) {                                                 /* :265-297 */
                    den = b0 * b2 - 1;
                    const double nw_ = (1 - w) * P[jm1] + w * (W * b2 - S) / den;
                    const double ns_ = (1 - w) * P[ip1] + w * (S * b0 - W) / den;
                    P[jm1] = nw_; P[ip1] = ns_;
                } else if (top) {                                                        /* :220-263 */
                    den = b0 * b2 * b3 - b0 - b2 - b3 - 2;
                    const double nw_ = (1 - w) * P[jm1] + w * (W * b2 * b3 - E * b2 - S * b3 - W - E - S) / den;
                    const double ns_ = (1 - w) * P[ip1] + w * (S * b0 * b3 - W * b3 + E * b0 - W + E - S) / den;
                    const double ne_ = (1 - w) * P[jp1] + w * (E * b0 * b2 - W * b2 + S * b0 - W - E + S) / den;
                    P[jm1] = nw_; P[ip1] = ns_; P[jp1] = ne_;
                } else if (bot && lef) {                                                 /* :518-541 */
                    den = b3 * b1 - 1;
                    const double nn_ = (1 - w) * P[im1] + w * (b3 * N - E) / den;
                    const double ne_ = (1 - w) * P[jp1] + w * (b1 * E - N) / den;
                    P[im1] = nn_; P[jp1] = ne_;
                } else if (bot && rig) {                                                 /* :589-613 */
                    den = b0 * b1 - 1;
                    const double nw_ = (1 - w) * P[jm1] + w * (W * b1 + N) / den;
                    const double nn_ = (1 - w) * P[im1] + w * (N * b0 + W) / den;
                    P[jm1] = nw_; P[im1] = nn_;
                } else if (bot) {                                                        /* :543-587 */
                    den = b0 * b1 * b3 - b0 - b1 - b3 - 2;
                    const double nw_ = (1 - w) * P[jm1] + w * (W * b1 * b3 - E + N - E * b1 - W + N * b3) / den;
                    const double nn_ = (1 - w) * P[im1] + w * (N * b0 * b3 + W - E - N - E * b0 + W * b3) / den;
                    const double ne_ = (1 - w) * P[jp1] + w * (E * b0 * b1 - N - W - W * b1 - N * b0 - E) / den;
                    P[jm1] = nw_; P[im1] = nn_; P[jp1] = ne_;
                } else if (lef) {                                                        /* :301-361 */
                    den = b1 * b2 * b3 - (b1 + b2 + b3) - 2;
                    const double nn_ = (1 - w) * P[im1] + w * (b2 * b3 * N - E * b2 - S * b3 - N - S - E) / den;
                    const double ns_ = (1 - w) * P[ip1] + w * (b1 * b3 * S + E * b1 - N * b3 - N - S + E) / den;
                    const double ne_ = (1 - w) * P[jp1] + w * (b1 * b2 * E - N * b2 + S * b1 - N + S - E) / den;
                    P[im1] = nn_; P[ip1] = ns_; P[jp1] = ne_;
                } else if (rig) {                                                        /* :468-514 */
                    den = (b0 * b1 * b2) + (-b0 - b1 - b2 - 2);
                    const double nw_ = (1 - w) * P[jm1] + w * (W * b1 * b2 - S + N - S * b1 - W + N * b2) / den;
                    const double nn_ = (1 - w) * P[im1] + w * (N * b0 * b2 + W - S - N - S * b0 + W * b2) / den;
                    const double ns_ = (1 - w) * P[ip1] + w * (S * b0 * b1 - N - W - W * b1 - N * b0 - S) / den;
                    P[jm1] = nw_; P[im1] = nn_; P[ip1] = ns_;
                } else {                                                                 /* inner cell, Gauss elimination :440-463 */
                    const double a = 1 / b0;
                    const double b = -(b0 + 1) / (b0 * b1 - 1);
                    const double alf = 1 + a;
                    const double gam = -a + b * alf;
                    const double x = N + a * W;
                    const double y = -a * W + b * x;
                    const double c = (1 - gam) / (b2 + gam);
                    P[jp1] = (1 - w) * P[jp1] + w * (E + y + c * (S + y)) / (b3 + gam + c * (gam - 1));
                    P[ip1] = (1 - w) * P[ip1] + w * (S + y + P[jp1] * (1 - gam)) / (b2 + gam);
                    P[im1] = (1 - w) * P[im1] + w * (x - alf * (P[jp1] + P[ip1])) / (b1 - a);
                    P[jm1] = (1 - w) * P[jm1] + w * (W + P[im1] - P[ip1] - P[jp1]) / (b0);
                }
            }
        for (int ci = 0, q = 0; ci < ny; ci++)                                           /* :616-640 */
            for (int cj = 0; cj < nx; cj++, q++) {
                const int i = 2 * ci + 1, j = 2 * cj + 1;
                u[q] = lambda * f[q] + lambda * (P[SG(i + 1, j)] - P[SG(i - 1, j)] + P[SG(i, j + 1)] - P[SG(i, j - 1)]);
                P1[q] = P[SG(i + 1, j)];
                P2[q] = P[SG(i, j + 1)];
            }
    }
    free(P); free(F); free(AL); free(ux); free(uy);
}
#undef SG

#define OCC_OMEGA 1.25             /* src/tvl1occflow_constants.h:28 */
/* src/tvl1occflow_solvers.cpp:150-216 Solver_wrt_u with the four dual planes as EXPLICIT in/out state (the reference
 * keeps them in function-local statics, zeroed when the width changes); n_iter = its MAX_ITERATIONS_U (10) */
void orc_occ_solver_u(double *u1, double *u2, const double *v1, const double *v2, const double *chi, const double *g,
                      double theta, double beta, int nx, int ny, double *p11, double *p12, double *p21, double *p22, int n_iter)
{
    const size_t n = (size_t) nx * ny;
    double *chix = dalloc(n), *chiy = dalloc(n), *f1 = dalloc(n), *f2 = dalloc(n);
    orc_forward_gradient(chi, chix, chiy, nx, ny);
    for (size_t i = 0; i < n; i++) {
        f1[i] = v1[i] / theta + beta * chix[i];
        f2[i] = v2[i] / theta + beta * chiy[i];
        u1[i] = v1[i] + theta * beta * chix[i];
        u2[i] = v2[i] + theta * beta * chiy[i];
    }
    orc_rof_box(u1, f1, p11, p12, g, theta, OCC_OMEGA, nx, ny, n_iter);
    orc_rof_box(u2, f2, p21, p22, g, theta, OCC_OMEGA, nx, ny, n_iter);
    free(chix); free(chiy); free(f1); free(f2);
}

/* ------------------------------------------------------------------------------------------ */
/* src/tvl1occflow.cpp: TV-L1 with occlusions, single scale (:144-329) and multiscale (:337-481), as the reference computes
 * it on a ZERO-FILLED heap: the dual planes of Solver_wrt_u and the dual variable of Solver_wrt_chi are function-local
 * statics there, re-created (zero) whenever the image width differs from the previous call's -- i.e. once per pyramid
 * level -- and kept across the warps and outer iterations of a level.  Here they are per-level arrays.
 * Constants: src/tvl1occflow_constants.h:26-41. */
#define OCC_EXT_MAX_ITERATIONS 20
#define OCC_MAX_ITERATIONS_CHI 100
#define OCC_MAX_ITERATIONS_U 10
#define OCC_G_FACTOR 0.05
#define OCC_TAU_ETA 0.15
#define OCC_TAU_CHI 0.15
#define OCC_PRESMOOTHING_SIGMA 0.8

/* iters (optional): outer iterations of every warp; state: 6 planes p11 p12 p21 p22 eta1 eta2 of this level */
void orc_tvl1occ_single_scale(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1,
                              double *u2, double *chi, int nx, int ny, double lambda, double alpha, double beta,
                              double theta, int warps, double epsilon, int verbose, double *state, int *iters)
{
    const size_t n = (size_t) nx * ny;
    double *buf = dalloc(26 * n), *q = buf;
    double *I1x = q, *I1y = (q += n), *I1w = (q += n), *I1wx = (q += n), *I1wy = (q += n), *I_1x = (q += n), *I_1y = (q += n);
    double *I_1w = (q += n), *I_1wx = (q += n), *I_1wy = (q += n), *rho1_c = (q += n), *rho3_c = (q += n), *v1 = (q += n);
    double *v2 = (q += n), *v11 = (q += n), *v12 = (q += n), *v31 = (q += n), *v32 = (q += n), *gp1 = (q += n), *gp2 = (q += n);
    double *grad1 = (q += n), *grad3 = (q += n), *g = (q += n), *u1prev = (q += n), *u2prev = (q += n), *Ix = (q += n);
    double *Iy = dalloc(n);
    orc_centered_gradient(filtI0, Ix, Iy, nx, ny);                                   /* choosed_g, choice 2, :96-133 */
    for (size_t i = 0; i < n; i++) {
        const double gggrad = sqrt(Ix[i] * Ix[i] + Iy[i] * Iy[i]);
        const double aux = 1. + OCC_G_FACTOR * gggrad;
        g[i] = 1. / aux;
    }
    orc_centered_gradient(I1, I1x, I1y, nx, ny);
    orc_centered_gradient(I_1, I_1x, I_1y, nx, ny);
    for (size_t i = 0; i < n; i++) {
        u1prev[i] = u1[i];
        u2prev[i] = u2[i];
        v1[i] = v2[i] = v11[i] = v12[i] = v31[i] = v32[i] = 0.0;
    }
    for (int w = 0; w < warps; w++) {
        orc_bicubic_warp(I1, u1, u2, I1w, nx, ny, 0);                                 /* border_out defaults to false */
        orc_bicubic_warp(I1x, u1, u2, I1wx, nx, ny, 0);
        orc_bicubic_warp(I1y, u1, u2, I1wy, nx, ny, 0);
        for (size_t i = 0; i < n; i++) { gp1[i] = -u1[i]; gp2[i] = -u2[i]; }
        orc_bicubic_warp(I_1, gp1, gp2, I_1w, nx, ny, 0);
        orc_bicubic_warp(I_1x, gp1, gp2, I_1wx, nx, ny, 0);
        orc_bicubic_warp(I_1y, gp1, gp2, I_1wy, nx, ny, 0);
        for (size_t i = 0; i < n; i++) {                                              /* :232-248 */
            grad1[i] = (I1wx[i] * I1wx[i] + I1wy[i] * I1wy[i]);
            grad3[i] = (I_1wx[i] * I_1wx[i] + I_1wy[i] * I_1wy[i]);
            rho1_c[i] = (I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i]);
            rho3_c[i] = (I_1w[i] + I_1wx[i] * u1[i] + I_1wy[i] * u2[i] - I0[i]);
        }
        int it = 0;
        double error = INFINITY;
        while (error > epsilon && it < OCC_EXT_MAX_ITERATIONS) {                      /* :255-277 */
            it++;
            orc_occ_solver_v(u1, u2, v1, v2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, v11, v12, v31, v32, grad1, grad3,
                             alpha, theta, lambda, nx, ny);
            orc_occ_solver_u(u1, u2, v1, v2, chi, g, theta, beta, nx, ny, state, state + n, state + 2 * n, state + 3 * n,
                             OCC_MAX_ITERATIONS_U);
            orc_median_filtering(u1, nx, ny, 3);
            orc_median_filtering(u2, nx, ny, 3);
            orc_occ_solver_chi(u1, u2, chi, I1wx, I1wy, I_1wx, I_1wy, rho1_c, rho3_c, v11, v12, v31, v32, g, lambda, theta,
                               alpha, beta, OCC_TAU_CHI, OCC_TAU_ETA, nx, ny, state + 4 * n, state + 5 * n, OCC_MAX_ITERATIONS_CHI);
            error = 0.0;                                                              /* L2error, :62-80 */
            for (size_t i = 0; i < n; i++) {
                error += (u1[i] - u1prev[i]) * (u1[i] - u1prev[i]) + (u2[i] - u2prev[i]) * (u2[i] - u2prev[i]);
                u1prev[i] = u1[i];
                u2prev[i] = u2[i];
            }
            error /= (int) n;
        }
        if (verbose) fprintf(stderr, "Warping: %d, Iterations: %d, Error: %e\n", w, it, error);
        if (iters) iters[w] = it;
    }
    free(buf);
    free(Iy);
}

/* returns 1 for "GaussianSmooth: sigma too large"; iters (optional): [scale][warp], scale 0 = finest */
int orc_tvl1occ_multiscale(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1, double *u2,
                           double *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta, int nscales,
                           double zfactor, int warps, double epsilon, int verbose, int *iters)
{
    if (nscales < 1 || nscales > 64) return 2;
    double *Im[64], *Ic[64], *Ip[64], *If[64], *U1[64], *U2[64], *CH[64];
    int nx[64], ny[64], rc = 0;
    const size_t size = (size_t) nxx * nyy;
    nx[0] = nxx; ny[0] = nyy;
    Im[0] = dalloc(size); Ic[0] = dalloc(size); Ip[0] = dalloc(size); If[0] = dalloc(size);
    U1[0] = u1; U2[0] = u2; CH[0] = chi;
    /* :382-395: image_normalization_4 is called, and its output immediately overwritten by the raw images */
    for (size_t i = 0; i < size; i++) {
        Im[0][i] = I_1[i]; Ic[0][i] = I0[i]; Ip[0][i] = I1[i]; If[0][i] = filtI0[i];
        u1[i] = u2[i] = chi[i] = 0.0;
    }
    rc |= orc_gaussian(Im[0], nxx, nyy, OCC_PRESMOOTHING_SIGMA);
    rc |= orc_gaussian(Ic[0], nxx, nyy, OCC_PRESMOOTHING_SIGMA);
    rc |= orc_gaussian(Ip[0], nxx, nyy, OCC_PRESMOOTHING_SIGMA);
    rc |= orc_gaussian(If[0], nxx, nyy, OCC_PRESMOOTHING_SIGMA);
    int built = 1;
    for (int s = 1; s < nscales && !rc; s++, built++) {
        orc_zoom_size(nx[s - 1], ny[s - 1], &nx[s], &ny[s], zfactor);
        const size_t sz = (size_t) nx[s] * ny[s];
        Im[s] = dalloc(sz); Ic[s] = dalloc(sz); Ip[s] = dalloc(sz); If[s] = dalloc(sz);
        U1[s] = dalloc(sz); U2[s] = dalloc(sz); CH[s] = dalloc(sz);
        for (size_t i = 0; i < sz; i++) U1[s][i] = U2[s][i] = CH[s][i] = 0.0;
        rc |= orc_zoom_out(Im[s - 1], Im[s], nx[s - 1], ny[s - 1], zfactor);          /* order of :421-424 */
        rc |= orc_zoom_out(Ic[s - 1], Ic[s], nx[s - 1], ny[s - 1], zfactor);
        rc |= orc_zoom_out(If[s - 1], If[s], nx[s - 1], ny[s - 1], zfactor);
        rc |= orc_zoom_out(Ip[s - 1], Ip[s], nx[s - 1], ny[s - 1], zfactor);
    }
    for (int s = nscales - 1; s >= 0 && !rc; s--) {
        const size_t sz = (size_t) nx[s] * ny[s];
        double *state = dalloc(6 * sz);
        for (size_t i = 0; i < 6 * sz; i++) state[i] = 0.0;
        orc_tvl1occ_single_scale(Im[s], Ic[s], Ip[s], If[s], U1[s], U2[s], CH[s], nx[s], ny[s], lambda, alpha, beta, theta,
                                 warps, epsilon, verbose, state, iters ? iters + s * warps : NULL);
        free(state);
        if (s) {
            orc_zoom_in(U1[s], U1[s - 1], nx[s], ny[s], nx[s - 1], ny[s - 1]);
            orc_zoom_in(U2[s], U2[s - 1], nx[s], ny[s], nx[s - 1], ny[s - 1]);
            orc_zoom_in(CH[s], CH[s - 1], nx[s], ny[s], nx[s - 1], ny[s - 1]);
            for (size_t i = 0; i < (size_t) nx[s - 1] * ny[s - 1]; i++) {
                U1[s - 1][i] *= (double) 1.0 / zfactor;
                U2[s - 1][i] *= (double) 1.0 / zfactor;
            }
        } else {
            for (size_t i = 0; i < sz; i++) CH[0][i] = (CH[0][i] > OCC_THR_CHI);     /* Threshold, :82-92 */
        }
    }
    for (int s = 0; s < built; s++) {
        free(Im[s]); free(Ic[s]); free(Ip[s]); free(If[s]);
        if (s) { free(U1[s]); free(U2[s]); free(CH[s]); }
    }
    return rc ? 1 : 0;
}
