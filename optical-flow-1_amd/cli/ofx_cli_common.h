/* ofx_cli_common.h -- shared by the three front-ends: context from the environment, flow -> .flo. */
#ifndef OFX_CLI_COMMON_H
#define OFX_CLI_COMMON_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ofx.h"
#include "ofx_io.h"

/* OFX_DEVICE = GPU index (default 0); OFX_PRECISION = f64 (default, strict) | f32 (float storage) */
static ofx_ctx *cli_context(void)
{
    const char *d = getenv("OFX_DEVICE"), *p = getenv("OFX_PRECISION");
    const int dev = d ? atoi(d) : 0;
    const int prec = (p && (!strcmp(p, "f32") || !strcmp(p, "F32"))) ? OFX_F32 : OFX_F64;
    ofx_ctx *ctx = NULL;
    const int s = ofx_ctx_create(&ctx, dev, prec);
    if (s != OFX_OK) {
        fprintf(stderr, "ERROR: cannot create a GPU context on device %d: %s\n", dev, ofx_strerror(s));
        return NULL;
    }
    return ctx;
}

/* cast to float, interleave, write (src/tvl1flow_main.cpp:209-214).  Only the .flo container is
 * implemented; the reference picks the format from the file suffix (src/iio.cpp:3671-3676). */
static int cli_save_flow(const char *outfile, const double *u, const double *v, int nx, int ny)
{
    if (!ofx_has_suffix(outfile, ".flo")) {
        fprintf(stderr, "ERROR: output \"%s\": only the .flo format is supported\n", outfile);
        return 1;
    }
    float *f = (float *) malloc(sizeof(float) * (size_t) nx * ny * 2);
    if (!f) return 1;
    for (size_t i = 0; i < (size_t) nx * ny; i++) {
        f[2 * i] = (float) u[i];
        f[2 * i + 1] = (float) v[i];
    }
    const int r = ofx_write_flo(outfile, f, nx, ny);
    free(f);
    if (r) fprintf(stderr, "ERROR: cannot write \"%s\"\n", outfile);
    return r;
}

#endif
