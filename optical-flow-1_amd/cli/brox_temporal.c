/* brox_temporal -- drop-in front-end for src/brox_temporal_main.cpp.
 *
 *   brox_temporal nimages I1...In [alpha gamma nscales zoom_factor TOL inner_iter outer_iter dir verbose]
 *
 * Writes dir/flow00.flo ... dir/flow(N-2).flo (:206-216).  Like the reference this program always exits with
 * status 0 (:229), clamps nscales so that the coarsest level keeps min(nx, ny) >= 16 (:180-185), and sends its
 * messages to stdout / stderr where the reference sends them.
 */
#include <math.h>

#include "ofx_cli_common.h"

/* src/brox_temporal_main.cpp:19-27 (defaults), :141-177 (ranges, silent) */
static const cli_opt OPTS[] = {
    {"alpha",       CLI_REAL, 18,     NULL, CLI_LE0, 0, NULL},
    {"gamma",       CLI_REAL, 7,      NULL, CLI_LT0, 0, NULL},
    {"nscales",     CLI_INT,  100,    NULL, CLI_LE0, 0, NULL},
    {"zoom_factor", CLI_REAL, 0.75,   NULL, CLI_LE0 | CLI_GE1, 0, NULL},
    {"TOL",         CLI_REAL, 0.0001, NULL, CLI_LE0, 0, NULL},
    {"inner_iter",  CLI_INT,  1,      NULL, CLI_LE0, 0, NULL},
    {"outer_iter",  CLI_INT,  15,     NULL, CLI_LE0, 0, NULL},
    {"dir",         CLI_TEXT, 0,      "./", CLI_ANY, 0, NULL},
    {"verbose",     CLI_INT,  0,      NULL, CLI_ANY, 0, NULL},
};
enum { O_ALPHA, O_GAMMA, O_NSCALES, O_ZFACTOR, O_TOL, O_INNER, O_OUTER, O_DIR, O_VERBOSE, O_COUNT };

int main(int argc, char *argv[])
{
    if (argc < 3) {
        printf("Usage: %s nimages I1...In [alpha gamma nscales zoom_factor TOL inner_iter outer_iter dir verbose]\n", argv[0]);
        return 0;
    }
    int i = 1;
    const int frames = atoi(argv[i]); i++;
    int nx = -1, ny = -1, correct = frames > 0 && argc >= 2 + frames;
    double *I = NULL;
    for (int f = 0; correct && f < frames; f++) {            /* read_images, :63-108 */
        int nxx, nyy;
        double *img = ofx_read_image_double(argv[i + f], &nxx, &nyy);
        correct = img != NULL;
        if (correct && f == 0) {
            nx = nxx; ny = nyy;
            I = (double *) malloc(sizeof(double) * (size_t) nx * ny * frames);
            correct = I != NULL;
        } else {
            correct = correct && nx == nxx && ny == nyy;
        }
        if (correct) memcpy(I + (size_t) f * nx * ny, img, sizeof(double) * (size_t) nx * ny);
        free(img);
    }
    i += frames > 0 ? frames : 0;
    cli_val o[O_COUNT];
    cli_parse(argc, argv, i, OPTS, O_COUNT, o);
    const double alpha = o[O_ALPHA].num, gamma = o[O_GAMMA].num, zfactor = o[O_ZFACTOR].num, TOL = o[O_TOL].num;
    const int initer = (int) o[O_INNER].num, outiter = (int) o[O_OUTER].num, verbose = (int) o[O_VERBOSE].num;
    int nscales = (int) o[O_NSCALES].num;
    const char *dir = o[O_DIR].text;

    if (!correct) {
        fprintf(stderr, "Cannot read the images or the size of the images are not equal\n");
        free(I);
        return 0;
    }
    const double N = 1 + log((nx < ny ? nx : ny) / 16.) / log(1. / zfactor);      /* :183-185 */
    if ((int) N < nscales) nscales = (int) N;
    if (verbose) {
        printf("\n alpha:%g gamma:%g scales:%d nu:%g TOL:%g inner:%d outer:%d\n", alpha, gamma, nscales, zfactor, TOL,
               initer, outiter);
        fflush(stdout);
    }
    if (frames <= 2) {                                       /* brox_optic_flow_temporal.cpp:537-541 */
        fprintf(stderr, "The method needs more than two frames\n");
        free(I);
        return 0;
    }
    ofx_ctx *ctx = cli_context();
    if (!ctx) { free(I); return 0; }
    const size_t n = (size_t) nx * ny;
    double *u = (double *) malloc(sizeof(double) * n * (frames - 1));
    double *v = (double *) malloc(sizeof(double) * n * (frames - 1));
    const int s = ofx_brox_temporal(ctx, I, u, v, nx, ny, frames, alpha, gamma, nscales, zfactor, TOL, initer, outiter,
                                    verbose);
    if (s != OFX_OK) {
        fprintf(stderr, "ERROR: %s (%s)\n", ofx_strerror(s), ofx_last_error(ctx));
    } else {
        for (int f = 0; f < frames - 1; f++) {              /* :206-216 */
            char file[4096];
            snprintf(file, sizeof(file), "%s/flow%.2d.flo", dir, f);
            if (cli_save_flow(file, u + f * n, v + f * n, nx, ny)) break;
        }
    }
    free(u); free(v); free(I);
    cli_write_stats(ctx, argv[0]);
    ofx_ctx_destroy(ctx);
    return 0;
}
