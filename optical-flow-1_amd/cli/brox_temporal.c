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

#define PAR_DEFAULT_ALPHA 18                /* src/brox_temporal_main.cpp:19-27 */
#define PAR_DEFAULT_GAMMA 7
#define PAR_DEFAULT_NSCALES 100
#define PAR_DEFAULT_ZFACTOR 0.75
#define PAR_DEFAULT_TOL 0.0001
#define PAR_DEFAULT_INNER_ITER 1
#define PAR_DEFAULT_OUTER_ITER 15
#define PAR_DEFAULT_DIR "./"
#define PAR_DEFAULT_VERBOSE 0

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
    double alpha   = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_ALPHA;      i++;
    double gamma   = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_GAMMA;      i++;
    int    nscales = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_NSCALES;    i++;
    double zfactor = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_ZFACTOR;    i++;
    double TOL     = (argc > i) ? atof(argv[i]) : PAR_DEFAULT_TOL;        i++;
    int    initer  = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_INNER_ITER; i++;
    int    outiter = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_OUTER_ITER; i++;
    const char *dir = (argc > i) ? argv[i] : PAR_DEFAULT_DIR;             i++;
    int    verbose = (argc > i) ? atoi(argv[i]) : PAR_DEFAULT_VERBOSE;    i++;

    if (alpha <= 0) alpha = PAR_DEFAULT_ALPHA;              /* :157-177 */
    if (gamma < 0) gamma = PAR_DEFAULT_GAMMA;
    if (nscales <= 0) nscales = PAR_DEFAULT_NSCALES;
    if (zfactor <= 0 || zfactor >= 1) zfactor = PAR_DEFAULT_ZFACTOR;
    if (TOL <= 0) TOL = PAR_DEFAULT_TOL;
    if (initer <= 0) initer = PAR_DEFAULT_INNER_ITER;
    if (outiter <= 0) outiter = PAR_DEFAULT_OUTER_ITER;

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
    ofx_ctx_destroy(ctx);
    return 0;
}
