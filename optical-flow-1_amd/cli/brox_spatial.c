/* brox_spatial -- drop-in front-end for src/brox_spatial_main.cpp.
 *
 *   brox_spatial I1 I2 [out_file nproc alpha gamma nscales zoom_factor TOL inner_iter outer_iter verbose]
 *
 * The `nproc` slot exists in the reference only when it is built with OpenMP (:102-104) -- which its
 * Makefile always does (src/Makefile:25) -- although the usage text omits it (:89-93).  It is accepted
 * and ignored here.  Like the reference this program ALWAYS exits with status 0 (:195), and its
 * messages go to stdout / std::cerr exactly where the reference sends them.
 */
#include <math.h>

#include "ofx_cli_common.h"

#define PAR_DEFAULT_NPROC 0                 /* src/brox_spatial_main.cpp:29-39 */
#define PAR_DEFAULT_ALPHA 50
#define PAR_DEFAULT_GAMMA 10
#define PAR_DEFAULT_NSCALES 10
#define PAR_DEFAULT_ZFACTOR 0.5
#define PAR_DEFAULT_TOL 0.0001
#define PAR_DEFAULT_INNER_ITER 1
#define PAR_DEFAULT_OUTER_ITER 15
#define PAR_DEFAULT_VERBOSE 0

int main(int argc, char *argv[])
{
    if (argc < 3) {
        printf("Usage: %s I1 I2 [out_file alpha gamma nscales zoom_factor TOL inner_iter outer_iter verbose]\n", argv[0]);
        return 0;
    }
    int i = 1;
    const char *image1 = argv[i]; i++;
    const char *image2 = argv[i]; i++;
    const char *outfile = (argc >= 4) ? argv[i] : "flow.flo"; i++;
    int    nproc   = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NPROC;      i++;
    double alpha   = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_ALPHA;      i++;
    double gamma   = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_GAMMA;      i++;
    int    nscales = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NSCALES;    i++;
    double zfactor = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_ZFACTOR;    i++;
    double TOL     = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_TOL;        i++;
    int    initer  = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_INNER_ITER; i++;
    int    outiter = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_OUTER_ITER; i++;
    int    verbose = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_VERBOSE;    i++;
    (void) nproc;

    if (alpha <= 0) alpha = PAR_DEFAULT_ALPHA;              /* :122-142 */
    if (gamma < 0) gamma = PAR_DEFAULT_GAMMA;
    if (nscales <= 0) nscales = PAR_DEFAULT_NSCALES;
    if (zfactor <= 0 || zfactor >= 1) zfactor = PAR_DEFAULT_ZFACTOR;
    if (TOL <= 0) TOL = PAR_DEFAULT_TOL;
    if (initer <= 0) initer = PAR_DEFAULT_INNER_ITER;
    if (outiter <= 0) outiter = PAR_DEFAULT_OUTER_ITER;

    int nx, ny, nx1, ny1;
    double *I1 = ofx_read_image_double(image1, &nx, &ny);
    double *I2 = ofx_read_image_double(image2, &nx1, &ny1);
    if (!I1 || !I2 || nx != nx1 || ny != ny1) {
        fprintf(stderr, "Cannot read the images or the size of the images are not equal\n");
        free(I1); free(I2);
        return 0;
    }
    /* Brox uses min(nx, ny), not the diagonal, and truncates N before comparing, :154-157 */
    const double N = 1 + log((nx < ny ? nx : ny) / 16.) / log(1. / zfactor);
    if ((int) N < nscales) nscales = (int) N;
    if (verbose) {
        printf("\n alpha:%g gamma:%g scales:%d nu:%g TOL:%g inner:%d outer:%d\n", alpha, gamma, nscales, zfactor, TOL,
               initer, outiter);
        fflush(stdout);
    }

    ofx_ctx *ctx = cli_context();
    if (!ctx) return 0;
    double *u = (double *) malloc(sizeof(double) * (size_t) nx * ny);
    double *v = (double *) malloc(sizeof(double) * (size_t) nx * ny);
    const int s = ofx_brox_spatial(ctx, I1, I2, u, v, nx, ny, alpha, gamma, nscales, zfactor, TOL, initer, outiter, verbose);
    if (s != OFX_OK) fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
    else (void) cli_save_flow(outfile, u, v, nx, ny);
    free(u); free(v); free(I1); free(I2);
    ofx_ctx_destroy(ctx);
    return 0;
}
