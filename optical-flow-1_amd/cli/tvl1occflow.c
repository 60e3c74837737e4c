/* tvl1occflow -- drop-in front-end of TV-L1 optical flow with occlusion detection: same positional arguments, defaults,
 * corrections, auto-nscales rule, verbose text and outputs (.flo flow, 0 / 255 occlusion map) as the reference's
 * src/tvl1occflow_main.cpp, with the solve done by libofx.so on the GPU (ofx_tvl1occ_multiscale).
 *
 *   tvl1occflow I_1 I0 I1 [I0_Smoothed out outOcc nproc lambda alpha beta theta nscales zfactor nwarps epsilon verbose]
 *
 * `nproc` is accepted and ignored (the reference passes it to omp_set_num_threads, :189-193).  The reference accepts two
 * image names and then dereferences the missing third (:88, :111); here that is a usage error.
 */
#include <math.h>

#include "ofx_cli_common.h"

/* src/tvl1occflow_constants.h:13-25 (defaults), src/tvl1occflow_main.cpp:126-186 (ranges; warnings unconditional but theta's) */
#define W CLI_WARN_ALWAYS
static const cli_opt OPTS[] = {
    {"I0_Smoothed", CLI_TEXT, 0, NULL, CLI_ANY, 0, NULL},                   /* default: the I0 argument */
    {"out",     CLI_TEXT, 0,    "flow.flo", CLI_ANY, 0, NULL},
    {"outOcc",  CLI_TEXT, 0,    "occlusions.png", CLI_ANY, 0, NULL},
    {"nproc",   CLI_INT,  1,    NULL, CLI_LT0 | W, 0, "warning: nproc changed to %d\n"},
    {"lambda",  CLI_REAL, 0.15, NULL, CLI_LE0 | W, 0, "warning: lambda changed to %g\n"},
    {"alpha",   CLI_REAL, 0.01, NULL, CLI_LE0 | W, 0, "warning: alpha changed to %g\n"},
    {"beta",    CLI_REAL, 0.15, NULL, CLI_LE0 | W, 0, "warning: beta changed to %g\n"},
    {"theta",   CLI_REAL, 0.3,  NULL, CLI_LE0, 0, "warning: theta changed to %g\n"},
    {"nscales", CLI_INT,  100,  NULL, CLI_LE0 | W, 0, "warning: nscales changed to %d\n"},
    {"zfactor", CLI_REAL, 0.5,  NULL, CLI_LE0 | CLI_GE1 | W, 0, "warning: zfactor changed to %g\n"},
    {"nwarps",  CLI_INT,  2,    NULL, CLI_LE0 | W, 0, "warning: nwarps changed to %d\n"},
    {"epsilon", CLI_REAL, 0.01, NULL, CLI_LE0 | W, 0, "warning: epsilon changed to %f\n"},
    {"verbose", CLI_INT,  0,    NULL, CLI_ANY, 0, NULL},
};
#undef W
enum { O_SMOOTH, O_OUT, O_OUTOCC, O_NPROC, O_LAMBDA, O_ALPHA, O_BETA, O_THETA, O_NSCALES, O_ZFACTOR, O_NWARPS, O_EPSILON,
       O_VERBOSE, O_COUNT };

int main(int argc, char *argv[])
{
    if (argc < 4) {
        fprintf(stderr, "Usage: %s I_1 I0 I1 [I0_Smoothed out outOcc nproc lambda alpha beta theta nscales zfactor nwarps epsilon "
                        "verbose  ]\n", *argv);
        return EXIT_FAILURE;
    }
    cli_val o[O_COUNT];
    cli_parse(argc, argv, 4, OPTS, O_COUNT, o);
    const char *names[4] = {argv[1], argv[2], argv[3], o[O_SMOOTH].text ? o[O_SMOOTH].text : argv[2]};
    const int nproc = (int) o[O_NPROC].num, nwarps = (int) o[O_NWARPS].num, verbose = (int) o[O_VERBOSE].num;
    int nscales = (int) o[O_NSCALES].num;
    const double lambda = o[O_LAMBDA].num, alpha = o[O_ALPHA].num, beta = o[O_BETA].num, theta = o[O_THETA].num;
    const double zfactor = o[O_ZFACTOR].num, epsilon = o[O_EPSILON].num;

    double *img[4] = {NULL, NULL, NULL, NULL};
    int nx[4], ny[4], bad = 0;
    for (int k = 0; k < 4; k++) {
        img[k] = ofx_read_image_double(names[k], &nx[k], &ny[k]);
        if (!img[k]) {
            fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", names[k]);     /* read_image, :36-52 */
            bad = 1;
        }
    }
    int rc = EXIT_SUCCESS;
    if (bad) rc = EXIT_FAILURE;
    /* images of different sizes: the reference computes nothing, writes nothing and exits with success (:203-204,:293) */
    else if (nx[0] == nx[1] && nx[2] == nx[1] && nx[3] == nx[1] && ny[0] == ny[1] && ny[2] == ny[1] && ny[3] == ny[1]) {
        const int w = nx[1], h = ny[1];
        /* no level smaller than 16x16, :209-213 (float ratio, then double logarithms) */
        const int N = floor(log((float) (w < h ? w : h) / 16.0) / log(1. / zfactor)) + 1;
        if (N < nscales) nscales = N;
        if (verbose)
            fprintf(stderr, " nproc=%d   \n lambda=%f \n alpha=%f \n beta=%f \n theta=%f \n nscales=%d \n zfactor=%f\n nwarps=%d \n"
                            " epsilon=%g\n", nproc, lambda, alpha, beta, theta, nscales, zfactor, nwarps, epsilon);
        ofx_ctx *ctx = cli_context();
        if (!ctx) return EXIT_FAILURE;
        const size_t n = (size_t) w * h;
        double *u = (double *) malloc(sizeof(double) * 3 * n);
        float *occ = (float *) malloc(sizeof(float) * n);
        double *v = u + n, *chi = u + 2 * n;
        const int s = ofx_tvl1occ_multiscale(ctx, img[0], img[1], img[2], img[3], u, v, chi, w, h, lambda, alpha, beta, theta,
                                             nscales, zfactor, nwarps, epsilon, verbose);
        if (s != OFX_OK) {
            /* e.g. an image smaller than 16 pixels (nscales < 1): the reference would index an empty pyramid */
            fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
            rc = EXIT_FAILURE;
        } else {
            if (cli_save_flow(o[O_OUT].text, u, v, w, h)) rc = EXIT_FAILURE;
            for (size_t i = 0; i < n; i++) occ[i] = (float) chi[i] * 255;                       /* :278-281 */
            const int r = ofx_write_gray_bytes(o[O_OUTOCC].text, occ, w, h);
            if (r) {
                fprintf(stderr, r == 2 ? "ERROR: occlusion map \"%s\": only PNG and PGM output are supported\n"
                                       : "ERROR: cannot write \"%s\"\n", o[O_OUTOCC].text);
                rc = EXIT_FAILURE;
            }
        }
        free(u); free(occ);
        cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    }
    for (int k = 0; k < 4; k++) free(img[k]);
    return rc;
}
