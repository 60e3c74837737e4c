/* horn_schunck_pyramidal -- drop-in front-end for src/horn_schunck_pyramidal_main.cpp.
 *
 *   horn_schunck_pyramidal I1 I2 [out_file processors alpha nscales zoom_factor nwarps TOL maxiter verbose]
 *
 * `processors` is accepted and ignored.  `maxiter` is not validated by the reference (:105) and is
 * passed through unchanged here too.
 */
#include <math.h>

#include "ofx_cli_common.h"

#define PAR_DEFAULT_NPROC 0                 /* src/horn_schunck_pyramidal_main.cpp:24-32 */
#define PAR_DEFAULT_ALPHA 7
#define PAR_DEFAULT_NSCALES 10
#define PAR_DEFAULT_ZFACTOR 0.5
#define PAR_DEFAULT_NWARPS 10
#define PAR_DEFAULT_TOL 0.0001
#define PAR_DEFAULT_MAXITER 150
#define PAR_DEFAULT_VERBOSE 0
#define PAR_MAX_ZFACTOR 0.99

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s I1 I2 [out_file processors alpha nscales zoom_factor nwarps TOL maxiter verbose]\n", *argv);
        return EXIT_FAILURE;
    }
    int i = 1;
    const char *image1 = argv[i]; i++;
    const char *image2 = argv[i]; i++;
    const char *outfile = (argc >= 4) ? argv[i] : "flow.flo"; i++;
    int    nproc   = (argc >= 5)  ? atoi(argv[i]) : PAR_DEFAULT_NPROC;   i++;
    double alpha   = (argc >= 6)  ? atof(argv[i]) : PAR_DEFAULT_ALPHA;   i++;
    int    nscales = (argc >= 7)  ? atoi(argv[i]) : PAR_DEFAULT_NSCALES; i++;
    double zfactor = (argc >= 8)  ? atof(argv[i]) : PAR_DEFAULT_ZFACTOR; i++;
    int    warps   = (argc >= 9)  ? atoi(argv[i]) : PAR_DEFAULT_NWARPS;  i++;
    double TOL     = (argc >= 10) ? atof(argv[i]) : PAR_DEFAULT_TOL;     i++;
    int    maxiter = (argc >= 11) ? atoi(argv[i]) : PAR_DEFAULT_MAXITER; i++;
    int    verbose = (argc >= 12) ? atoi(argv[i]) : PAR_DEFAULT_VERBOSE; i++;

    if (alpha <= 0) alpha = PAR_DEFAULT_ALPHA;              /* :101-118 */
    if (nscales <= 0) nscales = PAR_DEFAULT_NSCALES;
    if (zfactor <= 0) zfactor = PAR_DEFAULT_ZFACTOR;
    if (zfactor >= 1) zfactor = PAR_MAX_ZFACTOR;
    if (warps <= 0) warps = PAR_DEFAULT_NWARPS;
    if (TOL <= 0) TOL = PAR_DEFAULT_TOL;

    int nx, ny, nx1, ny1;
    double *I1 = ofx_read_image_double(image1, &nx, &ny);
    double *I2 = ofx_read_image_double(image2, &nx1, &ny1);
    if (!I1 || !I2 || nx != nx1 || ny != ny1) {
        fprintf(stderr, "Cannot read the input images or their sizes are different.\n");
        free(I1); free(I2);
        return EXIT_FAILURE;
    }
    const double N = 1 + log(hypot(nx, ny) / 16) / log(1 / zfactor);    /* :142-145 */
    if (N < nscales) nscales = (int) N;
    if (verbose)
        fprintf(stderr, "nproc=%d alpha=%g nscales=%d zfactor=%g warps=%d epsilon=%g\n", nproc, alpha, nscales, zfactor,
                warps, TOL);

    ofx_ctx *ctx = cli_context();
    if (!ctx) return EXIT_FAILURE;
    double *u = (double *) malloc(sizeof(double) * 2 * (size_t) nx * ny);
    double *v = u + (size_t) nx * ny;
    int rc = EXIT_SUCCESS;
    const int s = ofx_hs_pyramidal(ctx, I1, I2, u, v, nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter, verbose);
    if (s != OFX_OK) {
        fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
        rc = EXIT_FAILURE;
    } else if (cli_save_flow(outfile, u, v, nx, ny)) {
        rc = EXIT_FAILURE;
    }
    free(u); free(I1); free(I2);
    ofx_ctx_destroy(ctx);
    return rc;
}
