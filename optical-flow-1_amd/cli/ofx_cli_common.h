/* ofx_cli_common.h -- shared by the three front-ends: context from the environment, flow -> .flo. */
#ifndef OFX_CLI_COMMON_H
#define OFX_CLI_COMMON_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ofx.h"
#include "ofx_io.h"

/* OFX_DEVICE = GPU index (default 0); OFX_PRECISION = f64 (default, strict) | f32 (float storage); OFX_TOLERANCE=1: the f64
 * tolerance mode of the TV-L1 dual update (option relaxed_dual; not byte-identical .flo files); OFX_SOR_TOLERANCE=1: the tolerance
 * mode of the SOR solvers (option sor_exact = 0); OFX_STATS: see cli_write_stats */
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
    if (getenv("OFX_STATS")) ofx_set_option(ctx, "profile", 1);       /* HIP-event times of the iteration launches per scale */
    if (getenv("OFX_TOLERANCE")) ofx_set_option(ctx, "relaxed_dual", atoi(getenv("OFX_TOLERANCE")) != 0);
    /* OFX_SOR_TOLERANCE=1: Horn-Schunck / Brox with re-ordered sweeps inside the AEPE < 1e-4 bar (option sor_exact = 0) */
    if (getenv("OFX_SOR_TOLERANCE")) ofx_set_option(ctx, "sor_exact", atoi(getenv("OFX_SOR_TOLERANCE")) != 0 ? 0 : 1);
    return ctx;
}

/* OFX_STATS=path (or "-" for stderr): the work record of the solve as one JSON object -- what the reference only prints as
 * text when `verbose` (src/tvl1flow.cpp:184-188,284-286: scale sizes, iterations and error per warp), plus the kernel time
 * of the iteration launches per scale and the wall time of the call.  Nothing is written when the variable is unset. */
/* wall-clock phases of a front-end run (OFX_STATS only): cli_phase("name") closes the phase that began at the previous call (or
 * at the first call, which only starts the clock) */
#include <time.h>
#define CLI_MAX_PHASES 8
static struct { const char *name; double ms; } cli_phases[CLI_MAX_PHASES];
static int cli_nphases = 0;
static double cli_phase_t0 = -1.0;
__attribute__((unused)) static void cli_phase(const char *name)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    if (cli_phase_t0 >= 0.0 && name && cli_nphases < CLI_MAX_PHASES) {
        cli_phases[cli_nphases].name = name;
        cli_phases[cli_nphases++].ms = now - cli_phase_t0;
    }
    cli_phase_t0 = now;
}

__attribute__((unused)) static void cli_write_stats(const ofx_ctx *ctx, const char *program)
{
    const char *path = getenv("OFX_STATS");
    ofx_stats st;
    if (!path || !*path || ofx_get_stats(ctx, &st) != OFX_OK) return;
    FILE *f = strcmp(path, "-") ? fopen(path, "w") : stderr;
    if (!f) { fprintf(stderr, "warning: cannot write OFX_STATS file \"%s\"\n", path); return; }
    const char *base = strrchr(program, '/');
    fprintf(f, "{");
    if (cli_nphases) {
        fprintf(f, "\"phases_ms\": {");
        for (int k = 0; k < cli_nphases; k++) fprintf(f, "%s\"%s\": %.3f", k ? ", " : "", cli_phases[k].name, cli_phases[k].ms);
        fprintf(f, "}, ");
    }
    fprintf(f, "\"program\": \"%s\", \"nscales\": %d, \"solves_per_scale\": %d, \"work_pix_iters\": %.17g, \"total_ms\": %.6g, "
               "\"odd_stops\": %d, \"odd_stops_served_from_stored_state\": %d, \"scales\": [",
            base ? base + 1 : program, st.nscales, st.nsolves, st.work_pix_iters, st.total_ms, st.odd_stops, st.odd_stops_stored);
    const int ns = st.nscales < OFX_MAX_SCALES ? st.nscales : OFX_MAX_SCALES;
    const int nw = st.nsolves < OFX_MAX_SOLVES ? st.nsolves : OFX_MAX_SOLVES;
    for (int s = 0; s < ns; s++) {
        fprintf(f, "%s{\"scale\": %d, \"nx\": %d, \"ny\": %d, \"iterations\": [", s ? ", " : "", s, st.nx[s], st.ny[s]);
        for (int w = 0; w < nw; w++) fprintf(f, "%s%d", w ? ", " : "", st.iters[s][w]);
        fprintf(f, "], \"error\": [");
        for (int w = 0; w < nw; w++) fprintf(f, "%s%.17g", w ? ", " : "", st.error[s][w]);
        fprintf(f, "], \"iteration_kernel_ms\": %.6g}", st.iter_ms[s]);
    }
    fprintf(f, "]}\n");
    if (f != stderr) fclose(f);
}

/* ---- positional, optional, silently-corrected arguments ----------------------------------------------------------------
 * All the reference's front-ends read their options the same way: positional, every trailing one optional, a value
 * outside its range is replaced (by the default, or by a given substitute) instead of rejected -- with a warning on
 * stderr only in tvl1flow (when `verbose`, the LAST option, is set) and tvl1occflow (always) (src/tvl1flow_main.cpp:97-167,
 * src/horn_schunck_pyramidal_main.cpp:93-118, src/brox_spatial_main.cpp:102-142, src/brox_temporal_main.cpp:141-177).
 * One table per program (name, kind, default, range test, substitute, warning format) drives one parser. */
enum { CLI_INT, CLI_REAL, CLI_TEXT };
enum {                         /* when is the value out of range? */
    CLI_ANY = 0,               /* never checked (nproc of some programs, maxiter, verbose) */
    CLI_LE0 = 1,               /* value <= 0 */
    CLI_LT0 = 2,               /* value <  0 */
    CLI_GE1 = 4,               /* value >= 1            (may be combined with CLI_LE0) */
    CLI_GT_QUARTER = 8,        /* value > 0.25          (tvl1flow's tau) */
    CLI_WARN_ALWAYS = 16       /* the warning does not depend on `verbose` (tvl1occflow, all rows but theta) */
};
typedef struct {
    const char *name;
    int         kind;
    double      def;           /* default (numbers) */
    const char *def_text;      /* default (CLI_TEXT) */
    int         bad;           /* CLI_* range test */
    double      ge1_value;     /* substitute when only the CLI_GE1 test fires; 0 = the default */
    const char *warn;          /* printf format of the warning ("warning: tau changed to %g\n"), NULL = silent */
} cli_opt;
typedef struct {
    double      num;
    const char *text;
} cli_val;

/* argv[first ...] -> out[0 .. n-1]; then the range corrections, in table order, warnings to stderr iff out[n-1] (by
 * convention `verbose`) is non-zero and the row has a format */
__attribute__((unused)) static void cli_parse(int argc, char *argv[], int first, const cli_opt *opt, int n, cli_val *out)
{
    for (int k = 0; k < n; k++) {
        const char *a = (argc > first + k) ? argv[first + k] : NULL;
        out[k].text = a ? a : opt[k].def_text;
        out[k].num = opt[k].def;
        if (a && opt[k].kind == CLI_INT) out[k].num = atoi(a);
        if (a && opt[k].kind == CLI_REAL) out[k].num = atof(a);
    }
    const int verbose = n > 0 && out[n - 1].num != 0;
    for (int k = 0; k < n; k++) {
        const double v = out[k].num;
        const int low = ((opt[k].bad & CLI_LE0) && v <= 0) || ((opt[k].bad & CLI_LT0) && v < 0);
        const int high = ((opt[k].bad & CLI_GE1) && v >= 1) || ((opt[k].bad & CLI_GT_QUARTER) && v > 0.25);
        if (!low && !high) continue;
        out[k].num = (!low && high && opt[k].ge1_value != 0) ? opt[k].ge1_value : opt[k].def;
        if ((verbose || (opt[k].bad & CLI_WARN_ALWAYS)) && opt[k].warn) {
            if (opt[k].kind == CLI_INT) fprintf(stderr, opt[k].warn, (int) out[k].num);
            else fprintf(stderr, opt[k].warn, out[k].num);
        }
    }
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
