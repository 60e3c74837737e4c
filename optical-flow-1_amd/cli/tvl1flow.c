/* tvl1flow -- drop-in front-end: same positional arguments, defaults, clamping, auto-nscales rule,
 * verbose text and .flo output as the reference's src/tvl1flow_main.cpp, with the solve done by
 * libofx.so on the GPU.
 *
 *   tvl1flow I0 I1 [out nproc tau lambda theta nscales zfactor nwarps epsilon verbose]
 *
 * `nproc` is accepted and ignored (the reference passes it to omp_set_num_threads, :169-173).
 */
#include <math.h>

#include "ofx_cli_common.h"

#define PAR_DEFAULT_OUTFLOW "flow.flo"      /* src/tvl1flow_main.cpp:24-33 */
#define PAR_DEFAULT_NPROC   0
#define PAR_DEFAULT_TAU     0.25
#define PAR_DEFAULT_LAMBDA  0.15
#define PAR_DEFAULT_THETA   0.3
#define PAR_DEFAULT_NSCALES 100
#define PAR_DEFAULT_ZFACTOR 0.5
#define PAR_DEFAULT_NWARPS  5
#define PAR_DEFAULT_EPSILON 0.01
#define PAR_DEFAULT_VERBOSE 0

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s I0 I1 [out nproc tau lambda theta nscales zfactor nwarps epsilon verbose]\n", *argv);
        return EXIT_FAILURE;
    }
    int i = 1;
    const char *image1_name = argv[i]; i++;
    const char *image2_name = argv[i]; i++;
    const char *outfile = (argc > i) ? argv[i] : PAR_DEFAULT_OUTFLOW; i++;
    int    nproc   = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NPROC;   i++;
    double tau     = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_TAU;     i++;
    double lambda  = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_LAMBDA;  i++;
    double theta   = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_THETA;   i++;
    int    nscales = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NSCALES; i++;
    double zfactor = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_ZFACTOR; i++;
    int    nwarps  = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NWARPS;  i++;
    double epsilon = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_EPSILON; i++;
    int    verbose = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_VERBOSE; i++;

    /* out-of-range values silently fall back to the defaults, :102-167 */
    if (nproc < 0) { nproc = PAR_DEFAULT_NPROC; if (verbose) fprintf(stderr, "warning: nproc changed to %d\n", nproc); }
    if (tau <= 0 || tau > 0.25) { tau = PAR_DEFAULT_TAU; if (verbose) fprintf(stderr, "warning: tau changed to %g\n", tau); }
    if (lambda <= 0) { lambda = PAR_DEFAULT_LAMBDA; if (verbose) fprintf(stderr, "warning: lambda changed to %g\n", lambda); }
    if (theta <= 0) { theta = PAR_DEFAULT_THETA; if (verbose) fprintf(stderr, "warning: theta changed to %g\n", theta); }
    if (nscales <= 0) { nscales = PAR_DEFAULT_NSCALES; if (verbose) fprintf(stderr, "warning: nscales changed to %d\n", nscales); }
    if (zfactor <= 0 || zfactor >= 1) { zfactor = PAR_DEFAULT_ZFACTOR; if (verbose) fprintf(stderr, "warning: zfactor changed to %g\n", zfactor); }
    if (nwarps <= 0) { nwarps = PAR_DEFAULT_NWARPS; if (verbose) fprintf(stderr, "warning: nwarps changed to %d\n", nwarps); }
    if (epsilon <= 0) { epsilon = PAR_DEFAULT_EPSILON; if (verbose) fprintf(stderr, "warning: epsilon changed to %f\n", epsilon); }

    int nx, ny, nx2, ny2;
    double *I0 = ofx_read_image_double(image1_name, &nx, &ny);
    if (!I0) fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image1_name);
    double *I1 = ofx_read_image_double(image2_name, &nx2, &ny2);
    if (!I1) fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image2_name);
    if (!I0 || !I1) { free(I0); free(I1); return EXIT_FAILURE; }
    if (nx != nx2 || ny != ny2) {
        fprintf(stderr, "ERROR: input images size mismatch %dx%d != %dx%d\n", nx, ny, nx2, ny2);
        return EXIT_FAILURE;
    }

    /* smallest pyramid image not below ~16x16, :185-188 (N is truncated when assigned to the int) */
    const double N = 1 + log(hypot(nx, ny) / 16.0) / log(1 / zfactor);
    if (N < nscales) nscales = (int) N;
    if (verbose)
        fprintf(stderr, "nproc=%d tau=%f lambda=%f theta=%f nscales=%d zfactor=%f nwarps=%d epsilon=%g\n", nproc, tau,
                lambda, theta, nscales, zfactor, nwarps, epsilon);

    ofx_ctx *ctx = cli_context();
    if (!ctx) return EXIT_FAILURE;
    double *u = (double *) malloc(sizeof(double) * 2 * (size_t) nx * ny);
    double *v = u + (size_t) nx * ny;
    int rc = EXIT_SUCCESS;
    const int s = ofx_tvl1_multiscale(ctx, I0, I1, u, v, nx, ny, tau, lambda, theta, nscales, zfactor, nwarps, epsilon,
                                      verbose);
    if (s != OFX_OK) {
        /* the reference dies with an uncaught C++ exception here (e.g. "GaussianSmooth: sigma too large") */
        fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
        rc = EXIT_FAILURE;
    } else if (cli_save_flow(outfile, u, v, nx, ny)) {
        rc = EXIT_FAILURE;
    }
    free(u); free(I0); free(I1);
    ofx_ctx_destroy(ctx);
    return rc;
}
