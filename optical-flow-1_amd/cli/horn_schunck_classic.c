/* horn_schunck_classic -- drop-in front-end for src/horn_schunck_classic_main.cpp.
 *
 *   horn_schunck_classic niter alpha a b f
 *
 * Exactly 5 (or 6) arguments, otherwise the usage line goes to stderr and -- like the reference, which returns
 * fprintf's result (:25-27) -- the exit status is the number of characters printed (mod 256).  A size mismatch prints
 * "input images size mismatch" and returns likewise (:42-44).
 */
#include "ofx_cli_common.h"

int main(int argc, char *argv[])
{
    if (argc != 6 && argc != 7) return fprintf(stderr, "usage:\n\t%s niter alpha a b f\n", *argv);
    const int niter = atoi(argv[1]);
    const double alpha = atof(argv[2]);
    int w, h, ww, hh;
    double *a = ofx_read_image_double(argv[3], &w, &h);
    double *b = ofx_read_image_double(argv[4], &ww, &hh);
    if (!a || !b) {                        /* iio aborts with "could not read image" (iio.cpp); no flow is written */
        fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", a ? argv[4] : argv[3]);
        free(a); free(b);
        return EXIT_FAILURE;
    }
    if (w != ww || h != hh) return fprintf(stderr, "input images size mismatch\n");
    ofx_ctx *ctx = cli_context();
    if (!ctx) return EXIT_FAILURE;
    double *u = (double *) malloc(sizeof(double) * 2 * (size_t) w * h), *v = u + (size_t) w * h;
    const int s = ofx_hs_classic(ctx, a, b, u, v, w, h, niter, alpha);
    int rc = EXIT_SUCCESS;
    if (s != OFX_OK) {
        fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
        rc = EXIT_FAILURE;
    } else if (cli_save_flow(argv[5], u, v, w, h)) {
        rc = EXIT_FAILURE;
    }
    free(u); free(a); free(b);
    cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    return rc;
}
