/* tvl1flow -- drop-in front-end: same positional arguments, defaults, clamping, auto-nscales rule,
 * verbose text and .flo output as the reference's src/tvl1flow_main.cpp, with the solve done by
 * libofx.so on the GPU.
 *
 *   tvl1flow I0 I1 [out nproc tau lambda theta nscales zfactor nwarps epsilon verbose]
 *
 * `nproc` is accepted and ignored (the reference passes it to omp_set_num_threads, :169-173).
 */
#include <math.h>
#include <pthread.h>
#include <unistd.h>

#include "ofx_cli_common.h"

/* The HIP runtime needs ~0.2 s to initialise in a fresh process (measured: hipGetDeviceCount + the first hipStreamCreate,
 * profiles/r04_cli_budget_before.txt) -- two thirds of this program's wall time at 1080p.  It does not depend on the images, so the
 * context is created on a second thread while the main thread reads them. */
static void *context_thread(void *arg)
{
    *(ofx_ctx **) arg = cli_context();
    return NULL;
}

/* src/tvl1flow_main.cpp:24-33 (defaults), :97-167 (ranges; warnings only when verbose) */
static const cli_opt OPTS[] = {
    {"out",     CLI_TEXT, 0,    "flow.flo", CLI_ANY, 0, NULL},
    {"nproc",   CLI_INT,  0,    NULL, CLI_LT0, 0, "warning: nproc changed to %d\n"},
    {"tau",     CLI_REAL, 0.25, NULL, CLI_LE0 | CLI_GT_QUARTER, 0, "warning: tau changed to %g\n"},
    {"lambda",  CLI_REAL, 0.15, NULL, CLI_LE0, 0, "warning: lambda changed to %g\n"},
    {"theta",   CLI_REAL, 0.3,  NULL, CLI_LE0, 0, "warning: theta changed to %g\n"},
    {"nscales", CLI_INT,  100,  NULL, CLI_LE0, 0, "warning: nscales changed to %d\n"},
    {"zfactor", CLI_REAL, 0.5,  NULL, CLI_LE0 | CLI_GE1, 0, "warning: zfactor changed to %g\n"},
    {"nwarps",  CLI_INT,  5,    NULL, CLI_LE0, 0, "warning: nwarps changed to %d\n"},
    {"epsilon", CLI_REAL, 0.01, NULL, CLI_LE0, 0, "warning: epsilon changed to %f\n"},
    {"verbose", CLI_INT,  0,    NULL, CLI_ANY, 0, NULL},
};
enum { O_OUT, O_NPROC, O_TAU, O_LAMBDA, O_THETA, O_NSCALES, O_ZFACTOR, O_NWARPS, O_EPSILON, O_VERBOSE, O_COUNT };

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s I0 I1 [out nproc tau lambda theta nscales zfactor nwarps epsilon verbose]\n", *argv);
        return EXIT_FAILURE;
    }
    const char *image1_name = argv[1], *image2_name = argv[2];
    cli_phase(NULL);
    ofx_ctx *ctx = NULL;
    pthread_t ctx_thread;
    const int ctx_async = pthread_create(&ctx_thread, NULL, context_thread, &ctx) == 0;
    cli_val o[O_COUNT];
    cli_parse(argc, argv, 3, OPTS, O_COUNT, o);
    const char *outfile = o[O_OUT].text;
    const int nproc = (int) o[O_NPROC].num, nwarps = (int) o[O_NWARPS].num, verbose = (int) o[O_VERBOSE].num;
    int nscales = (int) o[O_NSCALES].num;
    const double tau = o[O_TAU].num, lambda = o[O_LAMBDA].num, theta = o[O_THETA].num, zfactor = o[O_ZFACTOR].num;
    const double epsilon = o[O_EPSILON].num;

    int nx, ny, nx2, ny2;
    double *I0 = ofx_read_image_double(image1_name, &nx, &ny);
    if (!I0) fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image1_name);
    double *I1 = ofx_read_image_double(image2_name, &nx2, &ny2);
    if (!I1) fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image2_name);
    if (!I0 || !I1 || nx != nx2 || ny != ny2) {
        if (I0 && I1) fprintf(stderr, "ERROR: input images size mismatch %dx%d != %dx%d\n", nx, ny, nx2, ny2);
        free(I0); free(I1);
        if (ctx_async) { pthread_join(ctx_thread, NULL); ofx_ctx_destroy(ctx); }
        return EXIT_FAILURE;
    }

    cli_phase("read_images");
    /* smallest pyramid image not below ~16x16, :185-188 (N is truncated when assigned to the int) */
    const double N = 1 + log(hypot(nx, ny) / 16.0) / log(1 / zfactor);
    if (N < nscales) nscales = (int) N;
    if (verbose)
        fprintf(stderr, "nproc=%d tau=%f lambda=%f theta=%f nscales=%d zfactor=%f nwarps=%d epsilon=%g\n", nproc, tau,
                lambda, theta, nscales, zfactor, nwarps, epsilon);

    if (ctx_async) pthread_join(ctx_thread, NULL);
    else ctx = cli_context();
    if (!ctx) return EXIT_FAILURE;
    cli_phase("wait_for_context");
    double *u = (double *) malloc(sizeof(double) * 2 * (size_t) nx * ny);
    double *v = u + (size_t) nx * ny;
    int rc = EXIT_SUCCESS;
    const int s = ofx_tvl1_multiscale(ctx, I0, I1, u, v, nx, ny, tau, lambda, theta, nscales, zfactor, nwarps, epsilon,
                                      verbose);
    cli_phase("solve");
    if (s != OFX_OK) {
        /* the reference dies with an uncaught C++ exception here (e.g. "GaussianSmooth: sigma too large") */
        fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
        rc = EXIT_FAILURE;
    } else if (cli_save_flow(outfile, u, v, nx, ny)) {
        rc = EXIT_FAILURE;
    }
    cli_phase("write_flo");
    free(u); free(I0); free(I1);
    cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    /* Everything this program owns is released and every file closed; what is left is the HIP runtime's own exit handlers, which
     * take 50-90 ms to tear down queues the kernel driver reclaims anyway.  OFX_FAST_EXIT=0 runs them. */
    const char *fe = getenv("OFX_FAST_EXIT");
    if (!fe || atoi(fe) != 0) {
        fflush(NULL);
        _exit(rc);
    }
    return rc;
}
