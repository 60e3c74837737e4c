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

/* src/brox_spatial_main.cpp:29-39 (defaults), :102-142 (ranges, silent) */
static const cli_opt OPTS[] = {
    {"out_file",    CLI_TEXT, 0,      "flow.flo", CLI_ANY, 0, NULL},
    {"nproc",       CLI_INT,  0,      NULL, CLI_ANY, 0, NULL},
    {"alpha",       CLI_REAL, 50,     NULL, CLI_LE0, 0, NULL},
    {"gamma",       CLI_REAL, 10,     NULL, CLI_LT0, 0, NULL},
    {"nscales",     CLI_INT,  10,     NULL, CLI_LE0, 0, NULL},
    {"zoom_factor", CLI_REAL, 0.5,    NULL, CLI_LE0 | CLI_GE1, 0, NULL},
    {"TOL",         CLI_REAL, 0.0001, NULL, CLI_LE0, 0, NULL},
    {"inner_iter",  CLI_INT,  1,      NULL, CLI_LE0, 0, NULL},
    {"outer_iter",  CLI_INT,  15,     NULL, CLI_LE0, 0, NULL},
    {"verbose",     CLI_INT,  0,      NULL, CLI_ANY, 0, NULL},
};
enum { O_OUT, O_NPROC, O_ALPHA, O_GAMMA, O_NSCALES, O_ZFACTOR, O_TOL, O_INNER, O_OUTER, O_VERBOSE, O_COUNT };

int main(int argc, char *argv[])
{
    if (argc < 3) {
        printf("Usage: %s I1 I2 [out_file alpha gamma nscales zoom_factor TOL inner_iter outer_iter verbose]\n", argv[0]);
        return 0;
    }
    const char *image1 = argv[1], *image2 = argv[2];
    cli_val o[O_COUNT];
    cli_parse(argc, argv, 3, OPTS, O_COUNT, o);
    const char *outfile = o[O_OUT].text;
    const int initer = (int) o[O_INNER].num, outiter = (int) o[O_OUTER].num, verbose = (int) o[O_VERBOSE].num;
    int nscales = (int) o[O_NSCALES].num;
    const double alpha = o[O_ALPHA].num, gamma = o[O_GAMMA].num, zfactor = o[O_ZFACTOR].num, TOL = o[O_TOL].num;

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
    cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    return 0;
}
