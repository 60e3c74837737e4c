/* horn_schunck_pyramidal -- drop-in front-end for src/horn_schunck_pyramidal_main.cpp.
 *
 *   horn_schunck_pyramidal I1 I2 [out_file processors alpha nscales zoom_factor nwarps TOL maxiter verbose]
 *
 * `processors` is accepted and ignored.  `maxiter` is not validated by the reference (:105) and is
 * passed through unchanged here too.
 */
#include <math.h>

#include "ofx_cli_common.h"

/* src/horn_schunck_pyramidal_main.cpp:24-32 (defaults), :93-118 (ranges, silent; zoom_factor >= 1 becomes 0.99) */
static const cli_opt OPTS[] = {
    {"out_file",    CLI_TEXT, 0,      "flow.flo", CLI_ANY, 0, NULL},
    {"processors",  CLI_INT,  0,      NULL, CLI_ANY, 0, NULL},
    {"alpha",       CLI_REAL, 7,      NULL, CLI_LE0, 0, NULL},
    {"nscales",     CLI_INT,  10,     NULL, CLI_LE0, 0, NULL},
    {"zoom_factor", CLI_REAL, 0.5,    NULL, CLI_LE0 | CLI_GE1, 0.99, NULL},
    {"nwarps",      CLI_INT,  10,     NULL, CLI_LE0, 0, NULL},
    {"TOL",         CLI_REAL, 0.0001, NULL, CLI_LE0, 0, NULL},
    {"maxiter",     CLI_INT,  150,    NULL, CLI_ANY, 0, NULL},
    {"verbose",     CLI_INT,  0,      NULL, CLI_ANY, 0, NULL},
};
enum { O_OUT, O_NPROC, O_ALPHA, O_NSCALES, O_ZFACTOR, O_NWARPS, O_TOL, O_MAXITER, O_VERBOSE, O_COUNT };

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s I1 I2 [out_file processors alpha nscales zoom_factor nwarps TOL maxiter verbose]\n", *argv);
        return EXIT_FAILURE;
    }
    const char *image1 = argv[1], *image2 = argv[2];
    cli_val o[O_COUNT];
    cli_parse(argc, argv, 3, OPTS, O_COUNT, o);
    const char *outfile = o[O_OUT].text;
    const int nproc = (int) o[O_NPROC].num, warps = (int) o[O_NWARPS].num, maxiter = (int) o[O_MAXITER].num;
    const int verbose = (int) o[O_VERBOSE].num;
    int nscales = (int) o[O_NSCALES].num;
    const double alpha = o[O_ALPHA].num, zfactor = o[O_ZFACTOR].num, TOL = o[O_TOL].num;

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
    cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    return rc;
}
