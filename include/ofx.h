/* include/ofx.h -- C ABI of libofx.so: the MI355X (gfx950) implementation of the variational
 * optical-flow hot path of 12334zq/optical-flow-1 (TV-L1 primal-dual solver + the shared stencil /
 * warp / pyramid operators; Horn-Schunck-pyramidal and Brox-spatial SOR solvers on top of them).
 *
 * The reference has no FFI / plugin layer: its boundary is the C++ library libof.a (mangled names,
 * `bool` and default arguments) called by its command-line programs.  Every entry
 * point below replaces one reference prototype 1:1 -- same argument order and meaning, with
 *   - a leading `ofx_ctx *` (one context = one GPU + one HIP stream + its workspace),
 *   - `bool` -> `int`,
 *   - an `int` status return instead of `void` + C++ exceptions (0 = OFX_OK).
 * Host-pointer entry points take/return dense row-major `double` arrays (the reference's ofpix_t,
 * src/of.h:4-10), index p = i*nx + j; device residency is internal.  The *_dev entry points take
 * device pointers (hipMalloc / torch tensors) and run on the context's stream.
 *
 * There is NO CPU fallback: without a usable gfx950 device ofx_ctx_create fails with
 * OFX_ERR_NODEV and every other call needs a context.
 *
 * Thread-safety: calls on different contexts are independent; one context must not be used from
 * two threads at once.  Calls are blocking (they return after the result is in the caller's
 * buffer), except where noted for *_dev.
 */
#ifndef OFX_H
#define OFX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_VERSION 100

/* status codes */
#define OFX_OK          0
#define OFX_ERR_ARG     1   /* bad size / NULL pointer / parameter out of range                  */
#define OFX_ERR_SIGMA   2   /* replaces std::runtime_error("GaussianSmooth: sigma too large"),
                               src/operators.cpp:520-522 (also returned where the reference would
                               read out of bounds: kernel radius >= image width or height)       */
#define OFX_ERR_NOMEM   3   /* replaces std::bad_alloc (device or host allocation failed)         */
#define OFX_ERR_HIP     4   /* a HIP runtime call or kernel launch failed                        */
#define OFX_ERR_NODEV   5   /* no gfx950 device / device index out of range                      */

/* storage precision of the device-resident arrays; arithmetic is ALWAYS double in registers.   */
#define OFX_F64 0           /* strict mode: double storage, per-pixel arithmetic bit-identical to
                               the reference (glibc-exact hypot, IEEE divisions, no FMA contraction) */
#define OFX_F32 1           /* fast mode: float storage (half the HBM traffic) and, in the TV-L1
                               dual update, hypot = sqrt(x*x+y*y) and one reciprocal per denominator;
                               AEPE vs the double reference ~1e-5, not bit-compatible             */

/* solver limits mirrored from the reference's #defines */
#define OFX_TVL1_MAX_ITERATIONS 300   /* src/tvl1flow.cpp:22  */
#define OFX_BROX_MAX_ITERATIONS 300   /* src/brox_optic_flow_spatial.cpp:24 */
#define OFX_HS_MAX_MAXITER 65536      /* largest Horn-Schunck `maxiter` accepted (reference: unbounded) */
#define OFX_MAX_SCALES 32             /* pyramid levels ofx_stats RECORDS; the solvers accept any number of levels
                                         (e.g. tvl1flow with zfactor = 0.9 at 1080p uses 47), levels >= 32 are solved
                                         but not itemised in the record (work_pix_iters still counts them)       */
#define OFX_MAX_SOLVES 64             /* warps (TV-L1, HS) or outer*inner solves (Brox) per scale */

typedef struct ofx_ctx ofx_ctx;

/* Work / timing record of the last solver call on a context (the reference only prints these on
 * stderr when `verbose`: src/tvl1flow.cpp:184-188,284-286). */
typedef struct ofx_stats {
    int    nscales;                                   /* pyramid levels actually used            */
    int    nsolves;                                   /* warps per scale (Brox: outer*inner)     */
    int    nx[OFX_MAX_SCALES], ny[OFX_MAX_SCALES];    /* level sizes, 0 = finest                 */
    int    iters[OFX_MAX_SCALES][OFX_MAX_SOLVES];     /* inner iterations / SOR sweeps per solve */
    double error[OFX_MAX_SCALES][OFX_MAX_SOLVES];     /* stopping-criterion value at exit        */
    double iter_ms[OFX_MAX_SCALES];                   /* HIP-event time of the inner-iteration
                                                         launches of a level (0 unless profiling) */
    long long iter_launches[OFX_MAX_SCALES];          /* iteration kernels that did real work    */
    double work_pix_iters;                            /* sum n_iter * nx_s * ny_s                */
    double total_ms;                                  /* wall time of the call on the host       */
    int    odd_stops;                                 /* TV-L1: loops that stopped on the first iteration of a
                                                         fused pair ...                                        */
    int    odd_stops_stored;                          /* ... of which the launch had stored its intermediate
                                                         state (option "store_a"): no recomputation needed     */
    int    fused[OFX_MAX_SCALES];                     /* TV-L1: iterations per iteration launch at each level
                                                         (1, 2, 3, or the tile kernel's 4 | 6); odd_stops then
                                                         counts the loops that ended inside such a launch unit  */
} ofx_stats;

/* ---- context ---------------------------------------------------------------------------------*/
int  ofx_device_count(void);                                   /* number of usable HIP devices    */
int  ofx_ctx_create(ofx_ctx **out, int device, int precision); /* precision: OFX_F64 | OFX_F32   */
void ofx_ctx_destroy(ofx_ctx *ctx);
const char *ofx_strerror(int status);
const char *ofx_last_error(const ofx_ctx *ctx);                /* detail text of the last failure */
void *ofx_ctx_stream(const ofx_ctx *ctx);                      /* the context's hipStream_t       */
int   ofx_ctx_precision(const ofx_ctx *ctx);
int   ofx_ctx_synchronize(ofx_ctx *ctx);
int   ofx_set_option(ofx_ctx *ctx, const char *name, double value);
/* options (value 0 = default / automatic unless noted):
 *   "profile"        0/1  bracket the inner-iteration launches with HIP events -> stats.iter_ms
 *   "fixed_work"     0/1  TV-L1: every warp runs exactly OFX_TVL1_MAX_ITERATIONS iterations (stopping
 *                         test disabled; the reference with epsilon = 0)
 *   "sor_exact"      HS / Brox: 1 (default) = the reference's sweep order, bit-identical results, K time steps
 *                         of the pipelined schedule per launch; 2 = the same schedule, one launch per time step;
 *                         0 = the TOLERANCE mode, re-ordered sweeps inside north_star's bar (AEPE < 1e-4 against the reference
 *                         on the BASELINE configs; csrc/ofx_sor_tile.hip): Horn-Schunck four-colour sweeps, K per launch on LDS
 *                         tiles (AEPE 9e-6 at config 3); Brox a checkerboard of 64 x 128-pixel tiles swept in the reference's
 *                         order inside a tile on the finest level and red-black sweeps below (AEPE 1.1e-5 at config 4; red-black
 *                         everywhere: 1.3e-4).  Lone solves 5-10x faster than exact, lockstep groups supported.  On inputs where
 *                         the solves run into maxiter unconverged (the discontinuous pair P1 at 1080p) ANY re-ordering moves the
 *                         flow by as much as the reference's own OpenMP threads do (AEPE 2.5e-2) -- use the exact mode there.
 *   "sor_fuse"       sor_exact = 0: sweeps per launch of the tile kernels, K = 1..4 (0 = default: Horn-Schunck 2 in lockstep groups of >= 4 pairs, else 4; Brox's red-black
 *                         levels 4); 9 = Brox's red-black levels through k_brox_sor, two launches per sweep (A/B); -1 = the round-1
 *                         kernels, one launch per colour and sweep (single pairs only; Brox: red-black on every level).  Results
 *                         do not depend on K.
 *   "sor_tile"       sor_exact = 0, Horn-Schunck: tile geometry 1 = 128 x 32 pixels on 16 waves, 2 = on 8 waves (default),
 *                         3 = 128 x 48 on 12 waves.  Results do not depend on it.
 *   "sor_wave_levels"  sor_exact = 0, Brox: pyramid levels 0 .. n - 1 use the checkerboard-of-tiles sweeps (default 1: the
 *                         finest), the coarser ones red-black.  PART OF THE RESULT (as is "sor_tile_w").
 *   "sor_tile_w"     ... columns per tile, 1..128 (default 128); "sor_wave_p": anti-diagonals of operand prefetch, 2 | 4 | 6 | 8
 *                         (default 4; results do not depend on it)
 *   "sor_batch"      sweeps in flight per batch in the exact modes (default 32 / 64)
 *   "sor_window"     time steps per launch of sor_exact = 1 (default 8; 4 for Brox in lockstep groups of >= 4 pairs)
 *   "sor_rows"       rows per workgroup (row block of a sweep) of sor_exact = 1 (default 64; 125 in lockstep groups
 *                         of >= 4 pairs)
 *   "sor_spw"        consecutive sweeps of a row block that share a workgroup in sor_exact = 1 (1, 2 or 4; default 0 = 1,
 *                         the fastest measured; results do not depend on it)
 *   "fuse2"          1/0  TV-L1: two iterations per kernel launch (default 1)
 *   "fuse3", "fuse3_min_px"  TV-L1: three iterations per launch (k_tvl1_iter3): 0 never, 1 on every level of at least
 *                         fuse3_min_px pixels x pairs, 2 (default) by measurement: not in strict mode, only for lockstep groups
 *                         or contexts sharing the device, levels of >= 500 000 pixels x pairs.  Results do not depend on it.
 *   "store_a"        TV-L1, fused pairs: a loop that stops on the first iteration of a pair needs the state between the
 *                         two iterations; 1 (default) = a launch also stores it when the previous error is within 1.5x
 *                         of the threshold (the host then just switches buffers), 0 = never (the iteration is
 *                         recomputed alone), 2 = always.  Results are bit-identical in all three settings.
 *   "nt_stores"      fused TV-L1 kernel: 0 (default) non-temporal stores when a launch's working set exceeds the Infinity
 *                    Cache, 1 always, 2 never (A/B measurements)
 *   "relaxed_dual"   0/1  TV-L1 in OFX_F64 storage: 1 = the tolerance mode -- double storage and arithmetic, but sqrt(x^2 + y^2)
 *                         for libm's hypot and reciprocals for the IEEE quotients of the dual update (one refinement step on
 *                         v_rsq_f64 / v_rcp_f64: relative error ~2^-45); NOT bit-identical: AEPE vs the reference ~1e-12 px on
 *                         the BASELINE configs (bar 1e-4), iteration tables equal there, 25-40 % faster than strict.
 *                         0 (default) = strict.  The front-ends read OFX_TOLERANCE=1 for it.
 *   "tile", "tile_max_px"  TV-L1: levels of at most tile_max_px pixels x pairs (default 200 000) run K = 4 | 6 iterations per
 *                         launch on 2-D tiles instead of the marching strips (0 = off, the default: measured no faster)
 *   "sor_lds"        windowed exact SOR sweeps: 0 = one global round trip per time step, 2 = the launch window staged in LDS,
 *                         1 (default) = by measurement (LDS for lone Horn-Schunck solves).  Results do not depend on it.
 *   "rof_pipe"       1/0  TV-L1 with occlusions / Scalar_ROF_BoxCellCentered: all iterations of a call in flight, the sweep
 *                         of iteration s 120 positions behind the sweep of s - 1 (default 1); 0 = one iteration at a time
 *   "rof_window"     steps per launch of the ROF box sweeps: 10 (default) or 24 (the round-2 geometry, 143 KB of LDS per workgroup)
 *   "chi_fuse"       1/0  Solver_wrt_chi: 5 iterations per launch on overlapping LDS tiles (default 1); 0 = two launches per
 *                         iteration.  Results do not depend on either.
 *   "gauss_fused"    pyramids of lockstep groups: 1 (default) row + column pass of the Gaussian in one launch, and for zfactor = 1/2
 *                         the whole zoom_out (smoothing + 2:1 sampling) in one; 3 = without the zoom_out fusion, 2 = the
 *                         generic-radius fused kernel, 0 = two passes.  Results do not depend on it.
 *   "spin_us"        microseconds the host spins on a convergence poll's pinned record before it sleeps in
 *                         hipEventSynchronize (default 150; 0 = never)
 *   "warp_lds"       1/0  TV-L1 warp with the bicubic taps staged through LDS (default 1)
 *   "lockstep"       pairs per lockstep group in ofx_tvl1_batch_dev (default 0 = ofx_tvl1_batch_group_size's
 *                         rule: as large as possible, evened out over the contexts; at most 16)
 *   "mem_budget"     bytes the level arrays of all contexts of a batch may occupy when the group size is chosen
 *                         (default 0 = half of the device memory that is free at the time of the call)
 *   "concurrency"    number of contexts that will be solving on the same device at the same time
 *                         (default 1); a scheduling hint for the strip height of the TV-L1 kernels
 *   "rows_per_wave", "rows_per_wave2", "chunk"   tuning of the TV-L1 kernels / launch batching */
int   ofx_get_stats(const ofx_ctx *ctx, ofx_stats *out);

/* ---- operators (replace src/operators.h:29-134) ---------------------------------------------*/
int ofx_divergence(ofx_ctx *ctx, const double *v1, const double *v2, double *div, int nx, int ny);
int ofx_forward_gradient(ofx_ctx *ctx, const double *f, double *fx, double *fy, int nx, int ny);
int ofx_centered_gradient(ofx_ctx *ctx, const double *f, double *dx, double *dy, int nx, int ny);  /* nz = 1 */
int ofx_dxx(ofx_ctx *ctx, const double *I, double *Ixx, int nx, int ny);                         /* nz = 1 */
int ofx_dyy(ofx_ctx *ctx, const double *I, double *Iyy, int nx, int ny);
int ofx_dxy(ofx_ctx *ctx, const double *I, double *Ixy, int nx, int ny);
/* in-place separable Gaussian with the reference's default arguments (reflecting boundary,
 * window 5): src/operators.cpp:506-624 */
int ofx_gaussian(ofx_ctx *ctx, double *I, int nx, int ny, double sigma);

/* centered_gradient3 (src/operators.h:107-114): f holds nz frames of nx*ny; dx, dy per frame, dz between frames */
int ofx_centered_gradient3(ofx_ctx *ctx, const double *f, double *dx, double *dy, double *dz, int nx, int ny, int nz);

/* ---- bicubic interpolation (replace src/bicubic_interpolation.h:16-52) -----------------------*/
/* n samples at (uu[k], vv[k]) -> out[k]; one call of the reference's bicubic_interpolation_at
 * per sample */
int ofx_bicubic_at(ofx_ctx *ctx, const double *input, const double *uu, const double *vv, double *out,
                   int n, int nx, int ny, int border_out);
int ofx_bicubic_warp(ofx_ctx *ctx, const double *input, const double *u, const double *v, double *output,
                     int nx, int ny, int border_out);
/* bicubic_interpolation_at_color (src/bicubic_interpolation.h:30-43): channel k of nz interleaved channels */
int ofx_bicubic_at_color(ofx_ctx *ctx, const double *input, const double *uu, const double *vv, double *out,
                         int n, int nx, int ny, int nz, int k, int border_out);

/* ---- pyramid zoom (replace src/zoom.h:20-63) -------------------------------------------------*/
void ofx_zoom_size(int nx, int ny, int *nxx, int *nyy, double factor);      /* pure host arithmetic */
int  ofx_zoom_out(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, double factor);
int  ofx_zoom_in(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, int nxx, int nyy);
/* zoom_out_color (src/zoom.h:44-55).  The reference is only defined for nz = 1 (for nz > 1 it reads beyond its
 * nx*ny scratch copy, src/zoom.cpp:96-118); nz = 1 is zoom_out, any other nz is OFX_ERR_ARG. */
int  ofx_zoom_out_color(ofx_ctx *ctx, const double *I, double *Iout, int nx, int ny, int nz, double factor);

/* ---- normalisation (replace src/utils.h:27-32) -----------------------------------------------*/
int ofx_image_normalization_2(ofx_ctx *ctx, const double *I1, const double *I2, double *I1n, double *I2n,
                              int size);
int ofx_image_normalization_1(ofx_ctx *ctx, const double *I, double *In, int size);      /* src/utils.h:17-20 */
int ofx_getminmax(ofx_ctx *ctx, const double *x, int n, double *min, double *max);        /* src/utils.h:119 */

/* ---- TV-L1 (replace src/tvl1flow.h:36-70) ----------------------------------------------------*/
/* single scale: u1/u2 are read as the initial flow and overwritten with the result */
int ofx_tvl1_single_scale(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2,
                          int nx, int ny, double tau, double lambda, double theta, int warps,
                          double epsilon, int verbose);
/* multiscale: incoming u1/u2 content is ignored (zeroed at the coarsest level) */
int ofx_tvl1_multiscale(ofx_ctx *ctx, const double *I0, const double *I1, double *u1, double *u2,
                        int nx, int ny, double tau, double lambda, double theta, int nscales,
                        double zfactor, int warps, double epsilon, int verbose);

/* Device-resident variant: dI0/dI1 are device arrays of nx*ny elements of the context's storage
 * precision (double for OFX_F64, float for OFX_F32); d_flo receives the Middlebury .flo payload,
 * nx*ny interleaved (u,v) float32 pairs (src/tvl1flow_main.cpp:209-213).  Runs on the context's
 * stream; returns after the solve has been fully enqueued AND its convergence tests resolved
 * (the result itself is complete once the stream is synchronised). */
int ofx_tvl1_multiscale_dev(ofx_ctx *ctx, const void *dI0, const void *dI1, void *d_flo,
                            int nx, int ny, double tau, double lambda, double theta, int nscales,
                            double zfactor, int warps, double epsilon, int verbose);

/* Lockstep group: n_pairs (1..16) independent pairs of the same size solved by ONE context with shared
 * kernel launches (every level array holds the pairs back to back, blockIdx.y = pair).  Each pair keeps
 * its own stopping test, iteration counts and ping-pong phase, so every flow is bit-identical to what
 * ofx_tvl1_multiscale_dev computes for that pair alone; what is shared is the launch latency of the small
 * pyramid levels and the host's convergence polls.  dI0/dI1/d_flo: n_pairs device pointers (layout of
 * ofx_tvl1_multiscale_dev); stats_out: optional, n_pairs records.  Asynchronous like the single-pair call. */
int ofx_tvl1_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI0, const void *const *dI1,
                       void *const *d_flo, int nx, int ny, double tau, double lambda, double theta,
                       int nscales, double zfactor, int warps, double epsilon, ofx_stats *stats_out);

/* Batch of independent pairs on one OR SEVERAL devices (the unit of parallel work is the image pair).
 * The pairs are cut into lockstep groups of ofx_tvl1_batch_group_size() consecutive pairs;
 * group q is solved on context ctxs[q % n_ctx] with ofx_tvl1_group_dev, one host thread per context, so
 * n_ctx groups are in flight at a time (each context = its own HIP stream and workspace; same precision).
 * The contexts may live on different GPUs: a group whose images or payload arrays are not device memory of
 * its context's GPU -- another GPU's memory, pinned or pageable host memory -- is staged through the context's
 * workspace (peer / host copies on the context's stream) and its payloads are copied back into the caller's
 * arrays: a single-process multi-GPU batch with the results gathered where the caller wants them.
 * Arrays dI0/dI1/d_flo hold n_pairs pointers with the layout of ofx_tvl1_multiscale_dev.  work_pix_iters (optional, n_pairs doubles)
 * receives sum n_iter*nx_s*ny_s per pair.  Returns after every pair has been fully solved (all
 * streams synchronised); the first failing group's status is returned. */
int ofx_tvl1_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI0, const void *const *dI1,
                       void *const *d_flo, int n_pairs, int nx, int ny, double tau, double lambda,
                       double theta, int nscales, double zfactor, int warps, double epsilon,
                       double *work_pix_iters);

/* Group size ofx_tvl1_batch_dev picks for this batch: the "lockstep" option of ctxs[0] if > 0, else the batch
 * is spread over the fewest rounds a group of 16 (less if device memory is short) allows, one group per
 * context and round, with the groups evened out.  Returns the size (>= 1), or -status (e.g. -OFX_ERR_ARG,
 * -OFX_ERR_SIGMA for a pyramid that cannot be built) on error. */
int ofx_tvl1_batch_group_size(ofx_ctx *const *ctxs, int n_ctx, int n_pairs, int nx, int ny, int nscales,
                              double zfactor);

/* Fixed-work inner loop only (src/tvl1flow.cpp:113-182 run exactly n_iter times on linearised
 * data, all arrays host double planes; u/p updated in place).  Returns the last error in *error.
 * Used by the kernel-level parity tests and the roofline measurement. */
int ofx_tvl1_iterations(ofx_ctx *ctx, double *u1, double *u2, double *p11, double *p12, double *p21,
                        double *p22, const double *I1wx, const double *I1wy, const double *rho_c,
                        int nx, int ny, double tau, double lambda, double theta, int n_iter,
                        double *error);

/* libm's hypot as the TV-L1 dual update calls it (src/tvl1flow.cpp:172-173), n independent evaluations on the device:
 * out[k] = hypot(x[k], y[k]) with the operation sequence of glibc >= 2.35, bit-identical to it over the whole
 * double range.  An operator entry for direct parity tests of the one libm function on the path. */
int ofx_hypot(ofx_ctx *ctx, const double *x, const double *y, double *out, int n);

/* How ofx_tvl1_batch_dev reaches a buffer on `mem_device` (-1 = host memory) from a context on `ctx_device`: 0 = in place,
 * 1 = one copy (host <-> device, or a peer copy when hipDeviceCanAccessPeer says yes), 2 = two copies through a pinned host
 * bounce buffer (the GPUs cannot address each other; the call succeeds and leaves a note in ofx_last_error).  A pure decision,
 * exported for the host-logic tests.  The cross-GPU path itself has never executed: no multi-GPU box was available in any round. */
int ofx_staging_route(int ctx_device, int mem_device, int can_access_peer);

/* ---- Horn-Schunck pyramidal (replace src/horn_schunck.h:15-48) --------------------------------*/
int ofx_hs_single_scale(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v,
                        int nx, int ny, double alpha, int warps, double TOL, int maxiter, int verbose);
int ofx_hs_pyramidal(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v,
                     int nx, int ny, double alpha, int nscales, double zfactor, int warps,
                     double TOL, int maxiter, int verbose);

/* Device-resident lockstep groups / batches of the SOR solvers, the counterparts of ofx_tvl1_group_dev and
 * ofx_tvl1_batch_dev (same pointer layout: n_pairs device images of the context's storage precision in, n_pairs
 * .flo payloads of nx*ny interleaved (u,v) float32 out).  The pairs of a group share every launch of the windowed
 * exact sweeps (blockIdx.z = pair) -- a lone solve is a latency chain of thousands of dependent steps with little
 * work each -- while every pair keeps its own error slots, snapshots, sweep counts and stopping test: each flow is
 * bit-identical to the one ofx_hs_pyramidal / ofx_brox_spatial compute for that pair alone.  Groups need the
 * default option sor_exact = 1 or the tile sweeps of sor_exact = 0.  The group size of the batch calls is option "lockstep" of ctxs[0] if > 0, else
 * as large as possible (<= 16), evened out over the contexts. */
int ofx_hs_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI1, const void *const *dI2, void *const *d_flo,
                     int nx, int ny, double alpha, int nscales, double zfactor, int warps, double TOL, int maxiter,
                     ofx_stats *stats_out);
int ofx_hs_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI1, const void *const *dI2,
                     void *const *d_flo, int n_pairs, int nx, int ny, double alpha, int nscales, double zfactor,
                     int warps, double TOL, int maxiter, double *work_pix_iters);

/* classic Horn-Schunck (replace src/horn_schunck.h:7-8, src/horn_schunck_classic.cpp:125-149): niter Jacobi
 * iterations from a zero flow; a / b are the two images, w x h */
int ofx_hs_classic(ofx_ctx *ctx, const double *a, const double *b, double *u, double *v, int w, int h, int niter,
                   double alpha);

/* ---- Brox spatial (replace src/brox_optic_flow.h:19-33) ---------------------------------------*/
int ofx_brox_spatial(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v,
                     int nxx, int nyy, double alpha, double gamma, int nscales, double nu,
                     double TOL, int inner_iter, int outer_iter, int verbose);

int ofx_brox_group_dev(ofx_ctx *ctx, int n_pairs, const void *const *dI1, const void *const *dI2, void *const *d_flo,
                       int nxx, int nyy, double alpha, double gamma, int nscales, double nu, double TOL,
                       int inner_iter, int outer_iter, ofx_stats *stats_out);
int ofx_brox_batch_dev(ofx_ctx *const *ctxs, int n_ctx, const void *const *dI1, const void *const *dI2,
                       void *const *d_flo, int n_pairs, int nxx, int nyy, double alpha, double gamma, int nscales,
                       double nu, double TOL, int inner_iter, int outer_iter, double *work_pix_iters);

/* ---- robust_expo_methods (replace src/robust_expo_methods.h:21-38) -----------------------------------*/
/* Brox's scheme with an image-driven weight in the smoothness term: method_type 1 = exp(-lambda |grad I1|), 2 = the same
 * + 0.001, 3 = lambda chosen per pixel from alpha and the gradient distribution.  The reference's argument order.  nzz must
 * be 1 (OFX_ERR_ARG otherwise): for colour images the reference's pyramid reads beyond its scratch copy (zoom.cpp:96-118).
 * Kept quirks of the source: the presmoothing is gaussian(I, nx, ny, nzz, 0.8), i.e. sigma = nzz = 1 with the DIRICHLET
 * boundary (robust_expo_methods.cpp:497-498), and alpha * nzz is truncated to an int (:529).  SOR sweeps run in the
 * reference's order (windowed exact schedule, option sor_exact = 1); bit-identical to the reference on one thread. */
int ofx_robust_expo(ofx_ctx *ctx, const double *I1, const double *I2, double *u, double *v, int nxx, int nyy, int nzz,
                    int method_type, double alpha, double gamma, double lambda, int nscales, double nu, double TOL,
                    int inner_iter, int outer_iter, int verbose);

/* ---- Brox temporal (replace src/brox_optic_flow.h:41-55) ---------------------------*/
/* I: `frames` images of nx*ny, frame-major; u, v: frames - 1 flow fields (u[f] takes frame f to frame f + 1).
 * frames <= 2 is an error ("The method needs more than two frames", brox_optic_flow_temporal.cpp:537-541: the
 * reference prints that and returns).  SOR sweeps run in the reference's order (windowed exact schedule). */
int ofx_brox_temporal(ofx_ctx *ctx, const double *I, double *u, double *v, int nxx, int nyy, int frames,
                      double alpha, double gamma, int nscales, double nu, double TOL, int inner_iter,
                      int outer_iter, int verbose);

/* ---- colour operators (replace src/bicubic_interpolation.h:66-75, src/utils.h:39-96) -----------------*/
/* bicubic_interpolation_warp_color: nz interleaved channels, every sample through bicubic_interpolation_at_color */
int ofx_bicubic_warp_color(ofx_ctx *ctx, const double *input, const double *u, const double *v, double *output,
                           int nx, int ny, int nz, int border_out);
/* image_normalization_2_color: per channel joint min / max of both images; `size` = number of ELEMENTS (pixels * nz,
 * a multiple of nz: the reference reads past its arrays otherwise), copy when max == min */
int ofx_image_normalization_2_color(ofx_ctx *ctx, const double *I1, const double *I2, double *I1n, double *I2n,
                                    int size, int nz);
/* image_normalization_3: joint min / max of three images, IN PLACE, no test for max == min (as the reference) */
int ofx_image_normalization_3(ofx_ctx *ctx, double *I0, double *I1, double *I2, int size);
int ofx_image_normalization_4(ofx_ctx *ctx, const double *I_1, const double *I0, const double *I1, const double *filtI0,
                              double *I_1n, double *I0n, double *I1n, double *filtI0n, int size);

/* ---- building blocks of TV-L1 with occlusions ---------------------------------------------------------------
 * The reference PROGRAM (tvl1occflow) is not reproducible by any implementation: Solver_wrt_chi reads its dual variable
 * uninitialised and Solver_wrt_u keeps its dual variables in function-local statics across calls
 * (src/tvl1occflow_solvers.cpp:161-186,239-263).  Its functions are deterministic once that state is explicit: */
/* me_median_filtering (src/utils.h:10, src/utils.cpp:150-213): wsize x wsize median, in place; wsize <= 9 */
int ofx_me_median_filtering(ofx_ctx *ctx, double *in, int nx, int ny, int wsize);
/* Solver_wrt_v (src/tvl1occflow_solvers.h, src/tvl1occflow_solvers.cpp:56-147); same argument order */
int ofx_solver_wrt_v(ofx_ctx *ctx, const double *u1, const double *u2, double *v1, double *v2, const double *chi,
                     const double *I1wx, const double *I1wy, const double *I_1wx, const double *I_1wy,
                     const double *rho1_c, const double *rho3_c, double *Vfwd_1, double *Vfwd_2, double *Vbck_1,
                     double *Vbck_2, const double *grad1, const double *grad3, double alpha, double theta, double lambda,
                     int nx, int ny);
/* Scalar_ROF_BoxCellCentered (src/tvl1occflow_tv_rof_box.h:52-55): nIter x { alfa from u; one in-place box-relaxation sweep in
 * the reference's cell order (executed on hyperplanes, bit-identical); u = lambda f + lambda div P }.  u: in = seed, out =
 * result; initialP1 / initialP2: in/out dual values on the south / east cell edges.  nx, ny >= 2. */
int ofx_scalar_rof_box_cell_centered(ofx_ctx *ctx, double *u, const double *f, double *initialP1, double *initialP2,
                                     const double *g_function, double lambda, double omega, int nx, int ny, int nIter);
/* Solver_wrt_u (src/tvl1occflow_solvers.cpp:150-216); the reference's argument order, followed by its four dual planes as
 * explicit in/out state (the reference keeps them in function-local statics, zeroed when the image width changes) and the
 * iteration count per flow component (the reference's MAX_ITERATIONS_U = 10) */
int ofx_solver_wrt_u(ofx_ctx *ctx, double *u1, double *u2, const double *v1, const double *v2, const double *chi,
                     const double *g, double theta, double beta, int nx, int ny, double *p11, double *p12, double *p21,
                     double *p22, int n_iter);
/* Solver_wrt_chi (src/tvl1occflow_solvers.cpp:218-337); the reference's argument order, followed by the dual variable
 * (eta1, eta2) as explicit in/out state -- zero it for what the reference computes on a zero-filled heap -- and the
 * iteration count (the reference's MAX_ITERATIONS_CHI = 100) */
int ofx_solver_wrt_chi(ofx_ctx *ctx, const double *u1, const double *u2, double *chi, const double *I1wx,
                       const double *I1wy, const double *I_1wx, const double *I_1wy, const double *rho1_c,
                       const double *rho3_c, const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1,
                       const double *Vbck_2, const double *g, double lambda, double theta, double alpha, double beta,
                       double tau_chi, double tau_eta, int nx, int ny, double *eta1, double *eta2, int n_iter);

/* The whole TV-L1-with-occlusions solve (replaces Dual_TVL1_optic_flow_multiscale, src/tvl1occflow.h / tvl1occflow.cpp:337-481;
 * the single-scale loop :144-329 runs on the device per level): I_1 / I0 / I1 = previous, current, next frame, filtI0 = the image
 * g = 1 / (1 + 0.05 |grad filtI0|) is taken from (the CLI passes I0 when no smoothed image is given); u1 / u2 / chi are outputs
 * (chi thresholded at 0.75 to {0, 1} as the reference does).  Host planes of nxx * nyy doubles; everything in between --
 * pyramids, warps, the three sub-solvers, the median, the stopping test -- stays on the device.  Double arithmetic and storage
 * whatever the context's precision.  The reference keeps the dual variables of Solver_wrt_u / Solver_wrt_chi in statics it
 * allocates uninitialised once per level; this entry point implements the zero-initialised reading.
 * ofx_get_stats: iters[scale][warp] = outer iterations, error = last L2 error. */
int ofx_tvl1occ_multiscale(ofx_ctx *ctx, const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1,
                           double *u2, double *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta,
                           int nscales, double zfactor, int warps, double epsilon, int verbose);
/* n_triples independent solves (e.g. the frames of a sequence), cut into lockstep groups of up to 16 consecutive triples
 * (option "lockstep" of ctxs[0]; fewer when the device memory -- option "mem_budget", default half of what is free -- does
 * not hold that many): the triples of a group share every kernel launch of a context, group q runs on context q mod n_ctx
 * (one host thread and one stream per context, all on one device).  Every result is bit-identical to the triple solved alone.
 * Arrays of n_triples host pointers.  Returns the first failing status (detail text on that context). */
int ofx_tvl1occ_batch(ofx_ctx *const *ctxs, int n_ctx, int n_triples, const double *const *I_1, const double *const *I0,
                      const double *const *I1, const double *const *filtI0, double *const *u1, double *const *u2,
                      double *const *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta, int nscales,
                      double zfactor, int warps, double epsilon);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H */
