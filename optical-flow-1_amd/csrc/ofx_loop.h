// ofx_loop.h -- data-dependent iteration loops without a host round trip per iteration.
//
// All three solvers have the same shape: `while (error > tol && n < max) { n++; sweep; error = f(sum) }`
// (src/tvl1flow.cpp:113, src/horn_schunck_pyramidal.cpp:143, src/brox_optic_flow_spatial.cpp:315).
// On the GPU every sweep k adds its per-wave partial sums into err[k][wave % 64] (one f64 atomic
// per wave).  The FIRST thing every kernel of sweep k+1 does is fetch the 64 shards of err[k]; it
// reduces them in a fixed butterfly order, applies the reference's own test and returns at once
// when the loop is over -- and then every later launch sees an untouched all-zero slot and is a
// no-op too.  The host enqueues sweeps in chunks; a one-block finalize kernel after each chunk
// publishes {n, done, error} into pinned host memory, and the host reads that record one chunk
// behind the GPU.  `n` is therefore exactly the reference's iteration count.
#pragma once

#include "ofx_internal.h"
#include "ofx_device.h"

#define OFX_CRIT_MEAN      0   // error = sum / size           (TV-L1, tvl1flow.cpp:162)
#define OFX_CRIT_SQRT_MEAN 1   // error = sqrt(sum / size)     (HS :230, Brox :389)

OFX_DEV double loop_error_from_sum(double sum, int size, int crit)
{
    const double m = sum / size;
    return crit == OFX_CRIT_SQRT_MEAN ? sqrt(m) : m;
}

// Fetch this lane's shard of sweep k-1 (call first, use late: the load overlaps with other loads).
OFX_DEV double loop_fetch_prev(const double *err, int k)
{
    return k > 0 ? err[(size_t) (k - 1) * OFX_NSHARD + (threadIdx.x & 63)] : 0.0;
}

// true when sweep k must run.  Full waves only.
OFX_DEV bool loop_continues(double prev_shard, int k, int size, double thr, int crit)
{
    if (k == 0) return true;
    const double error = loop_error_from_sum(wave_allreduce_sum(prev_shard), size, crit);
    return error > thr;
}

// Add a wave's partial sum into sweep k's slot.  Full waves only; `wave_id` spreads the shards.
OFX_DEV void loop_accumulate(double *err, int k, double lane_value, int wave_id)
{
    const double s = wave_allreduce_sum(lane_value);
    if ((threadIdx.x & 63) == 0) atomicAdd(err + (size_t) k * OFX_NSHARD + (wave_id & (OFX_NSHARD - 1)), s);
}

struct LoopSpec {
    int    max_iter;   // reference's MAX_ITERATIONS / maxiter
    int    size;       // nx * ny of the level
    double thr;        // eps^2 (TV-L1) or TOL (SOR)
    int    crit;       // OFX_CRIT_*
    int    chunk;      // sweeps per poll
    bool   fixed;      // run exactly max_iter sweeps (stopping test disabled)
    bool   pairs;      // the solver fuses two sweeps per launch where it can (TV-L1)
    int    fuse = 0;   // > 2: the solver fuses this many sweeps per launch (TV-L1 tile kernel of the small levels); a loop that
                       // stops inside a launch unit is finished by redo(), which re-runs the unit's first sweeps from its input
    double afac = 0.0; // pairs only: a fused launch whose previous error is <= thr * afac also stores the state between
                       // its two sweeps (0 = never); the finalize kernel reports whether the stopping launch did
};

int ofx_loop_reserve(ofx_ctx *ctx, int max_iter);                      // err slots for max_iter sweeps
int ofx_loop_clear(ofx_ctx *ctx, size_t slots);                        // zero the loop state + the first `slots` error slots
// G problems in lockstep: problem g owns err slots [g * slots_per_problem, ...) and state entry g.
// seq: number the records carry when they are complete (never 0); ofx_loop_wait_poll(slot, G, seq) waits for them
int ofx_loop_finalize_group(ofx_ctx *ctx, const LoopSpec &L, int G, int slots_per_problem, int start, int launched,
                            OfxIterState *host_slot, int seq);
int ofx_loop_finalize_cursor(ofx_ctx *ctx, const LoopSpec &L, int G, int slots_per_problem, int launch_end, int max_range,
                             OfxIterState *host_slot, int seq);
int ofx_loop_wait_poll(ofx_ctx *ctx, int slot, int G, int seq);
static inline int ofx_poll_seq(const ofx_ctx *ctx) { return (int) (ctx->poll_seq & 0x3FFFFFFF) + 1; }

// Runs the loop for G independent problems of the same size in LOCKSTEP (TV-L1: G image pairs solved by
// the same launches, blockIdx.y = problem; the SOR solvers use G = 1).  Problem g accumulates into err slots
// g * (L.max_iter + 1) + k and stops on its own test -- its launches turn into no-ops while the others
// continue; the host stops enqueueing when every problem is done.
//   launch(k, count, thr) must enqueue every kernel of sweeps k .. k+count-1 of ALL problems on ctx->stream
// (count is 2 only when L.pairs and at least two sweeps remain; pairs always start at an even k); kernels
// take (ctx->d_err, k, thr) and use the helpers above (L.fuse > 2: count is up to L.fuse, units start at multiples of it, and
// redo(k_of) must re-run the first n_g - k0 sweeps of the unit that contains sweep k_of[g] = n_g - 1 from that unit's input).
// With L.pairs the second sweep of a pair runs even
// when the first one ended the loop; if that happens for any problem (n odd), redo(k_of) is called ONCE with
// k_of[g] = n_g - 1 for those problems and -1 for the others, and must recompute sweep k_of[g] of each such
// problem alone from the pair's untouched input buffers.  Problems whose stopping launch had stored its intermediate
// state (L.afac, or bit g of amask0 for a loop that stopped in its very first launch) need no redo: took_alt[g] = 1
// tells the caller to continue from that stored state.  Returns the reference's n and error per problem.
template <class LaunchFn, class RedoFn>
static int ofx_run_loop_group_impl(ofx_ctx *ctx, const LoopSpec &L, int G, LaunchFn launch, RedoFn redo, int *n_out,
                                   double *err_out, float *ms_out, unsigned amask0, int *took_alt);

// An error return must not leave launches or a poll of this loop in flight: the next API call resets the arena these
// kernels work in.  (Stream order already protects the next call's own kernels; the drain makes it hold for anything
// else the caller does with the context.)
template <class LaunchFn, class RedoFn>
static int ofx_run_loop_group(ofx_ctx *ctx, const LoopSpec &L, int G, LaunchFn launch, RedoFn redo, int *n_out,
                              double *err_out, float *ms_out, unsigned amask0 = 0, int *took_alt = nullptr)
{
    const int s = ofx_run_loop_group_impl(ctx, L, G, launch, redo, n_out, err_out, ms_out, amask0, took_alt);
    if (s != OFX_OK) (void) hipStreamSynchronize(ctx->stream);
    return s;
}

template <class LaunchFn, class RedoFn>
static int ofx_run_loop_group_impl(ofx_ctx *ctx, const LoopSpec &L, int G, LaunchFn launch, RedoFn redo, int *n_out,
                                   double *err_out, float *ms_out, unsigned amask0, int *took_alt)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "loop group of %d problems", G);
    const int per = L.max_iter + 1;                      // +1: scratch slot for the redo's error
    OFX_TRY(ofx_loop_reserve(ctx, G * per));
    LoopSpec S = L;
    if (S.fixed) S.thr = -1.0;          // error >= 0 always passes `error > -1`
    OFX_TRY(ofx_loop_clear(ctx, (size_t) G * per));
    if (ms_out) OFX_HIP(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));

    // head / tail count the polls issued / consumed.  A context that has the device to itself keeps two
    // outstanding (the one the host waits for and one chunk of lookahead, so the GPU never idles while the
    // host reads a poll) at the price of one chunk of no-op launches behind the stopping iteration.  When
    // other contexts share the device (option "concurrency" > 1) their work fills such gaps, so nothing is
    // launched speculatively: one outstanding poll.
    const int max_out = ctx->concurrency > 1 ? 1 : 2;
    int launched = 0, head = 0, tail = 0, slot_of[2] = {0, 0}, seq_of[2] = {0, 0};
    bool stop = false;
    OfxIterState fin[OFX_MAX_GROUP];
    int chunk = S.chunk < 1 ? 1 : S.chunk;
    const int F = S.fuse > 2 ? S.fuse : (S.pairs ? 2 : 1);       // sweeps per launch unit; units start at multiples of F
    chunk = (chunk + F - 1) / F * F;
    for (;;) {
        while (launched < S.max_iter && head - tail < max_out) {
            const int first = launched;
            const int c = (S.max_iter - launched < chunk) ? S.max_iter - launched : chunk;
            while (launched < first + c) {
                const int cnt = (first + c - launched < F) ? first + c - launched : F;
                OFX_TRY(launch(launched, cnt, S.thr));
                launched += cnt;
            }
            const int seq = ofx_poll_seq(ctx);
            const int slot = (int) (ctx->poll_seq++ % OFX_NPOLL);
            OFX_TRY(ofx_loop_finalize_group(ctx, S, G, per, first, launched, &ctx->h_state[slot * OFX_MAX_GROUP], seq));
            slot_of[head & 1] = slot;
            seq_of[head & 1] = seq;
            OFX_HIP(ctx, hipEventRecord(ctx->ev_poll[slot], ctx->stream));
            head++;
        }
        if (tail == head) break;
        const int slot = slot_of[tail & 1];
        OFX_TRY(ofx_loop_wait_poll(ctx, slot, G, seq_of[tail & 1]));
        bool all = true;
        for (int g = 0; g < G; g++) {
            fin[g] = ctx->h_state[slot * OFX_MAX_GROUP + g];
            all = all && fin[g].done;
        }
        tail++;
        if (all) {
            // A poll still in flight covers launches that are no-ops (their stopping test already
            // fails); it is not drained -- stream order keeps it ahead of whatever is enqueued next.
            stop = true;
            break;
        }
    }
    if (!stop) return ofx_fail(ctx, OFX_ERR_HIP, "iteration loop ended without a final state");
    int redo_k[OFX_MAX_GROUP];
    bool any_redo = false;
    for (int g = 0; g < G; g++) {
        // the loop ended inside launch unit j = (n - 1) / F unless n is that unit's last sweep (units are F sweeps, the last
        // one of the loop what is left of max_iter): F = 2 -> n odd and not max_iter
        bool r = false;
        if (F > 1 && fin[g].n > 0) {
            const int k0 = (fin[g].n - 1) / F * F, cnt = (S.max_iter - k0 < F) ? S.max_iter - k0 : F;
            r = fin[g].n - k0 != cnt;
        }
        const bool stored = r && F == 2 && took_alt && (fin[g].apred || (fin[g].n == 1 && ((amask0 >> g) & 1u)));
        if (took_alt) took_alt[g] = stored ? 1 : 0;
        r = r && !stored;
        redo_k[g] = r ? fin[g].n - 1 : -1;
        any_redo = any_redo || r;
    }
    if (any_redo) OFX_TRY(redo(redo_k));
    if (ms_out) OFX_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
    for (int g = 0; g < G; g++) {
        n_out[g] = fin[g].n;
        err_out[g] = fin[g].error;
    }
    if (ms_out) {
        OFX_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
        OFX_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev_t0, ctx->ev_t1));
    }
    return OFX_OK;
}

// single problem (the SOR solvers); redo(k) as above for the one problem
template <class LaunchFn, class RedoFn>
static int ofx_run_loop(ofx_ctx *ctx, const LoopSpec &L, LaunchFn launch, RedoFn redo, int *n_out, double *err_out,
                        float *ms_out)
{
    return ofx_run_loop_group(ctx, L, 1, launch, [&](const int *k_of) { return redo(k_of[0]); }, n_out, err_out, ms_out);
}


// ---- cursor loops ---------------------------------------------------------------------------------------------------------
// The launch units of a loop have NO fixed size: every launch reads where its pair stands from device memory (OfxLoopDev), picks
// its own iteration count (TV-L1: three iterations while the loop is far from its threshold, fewer when the previous error says
// the stop is near -- fewer iterations computed past the stop, fewer re-runs) and logs where it started.  The host only counts
// launches: launch(L, thr) enqueues launch L for all problems; units_per_chunk launches, then one finalize kernel (which learns
// the iteration range from the device state), polled one chunk behind as in ofx_run_loop_group.  On return unit_out[g] / k0_out[g] /
// ucnt_out[g] describe the launch unit that contains iteration n_g - 1; when n_g - k0 < ucnt the loop ended INSIDE that unit and
// redo(n_of, k0_of, unit_of) -- called once, entries -1 for the problems that need nothing -- must re-run its first n - k0
// iterations from the unit's input buffers.
template <class LaunchFn, class RedoFn>
static int ofx_run_loop_cursor(ofx_ctx *ctx, const LoopSpec &L, int G, int units_per_chunk, int max_unit, LaunchFn launch, RedoFn redo,
                               int *n_out, double *err_out, int *unit_out, int *inside_out, float *ms_out)
{
    if (G < 1 || G > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "loop group of %d problems", G);
    const int per = L.max_iter + 1;
    OFX_TRY(ofx_loop_reserve(ctx, G * per));
    LoopSpec S = L;
    if (S.fixed) S.thr = -1.0;
    OFX_TRY(ofx_loop_clear(ctx, (size_t) G * per));
    if (ms_out) OFX_HIP(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));
    const int max_out = ctx->concurrency > 1 ? 1 : 2;
    const int upc = units_per_chunk < 1 ? 1 : units_per_chunk;
    int launches = 0, head = 0, tail = 0, slot_of[2] = {0, 0}, seq_of[2] = {0, 0};
    bool stop = false;
    OfxIterState fin[OFX_MAX_GROUP];
    int status = OFX_OK;
    for (;;) {
        while (launches < OFX_ULOG && head - tail < max_out) {
            const int c = OFX_ULOG - launches < upc ? OFX_ULOG - launches : upc;
            for (int i = 0; i < c && status == OFX_OK; i++) status = launch(launches++, S.thr);
            if (status != OFX_OK) break;
            const int seq = ofx_poll_seq(ctx);
            const int slot = (int) (ctx->poll_seq++ % OFX_NPOLL);
            status = ofx_loop_finalize_cursor(ctx, S, G, per, launches, c * max_unit, &ctx->h_state[slot * OFX_MAX_GROUP], seq);
            if (status != OFX_OK) break;
            slot_of[head & 1] = slot;
            seq_of[head & 1] = seq;
            if (hipEventRecord(ctx->ev_poll[slot], ctx->stream) != hipSuccess) { status = ofx_fail(ctx, OFX_ERR_HIP, "event record failed"); break; }
            head++;
        }
        if (status != OFX_OK || tail == head) break;
        const int slot = slot_of[tail & 1];
        status = ofx_loop_wait_poll(ctx, slot, G, seq_of[tail & 1]);
        if (status != OFX_OK) break;
        bool all = true;
        for (int g = 0; g < G; g++) {
            fin[g] = ctx->h_state[slot * OFX_MAX_GROUP + g];
            all = all && fin[g].done;
        }
        tail++;
        if (all) { stop = true; break; }
    }
    if (status == OFX_OK && !stop) status = ofx_fail(ctx, OFX_ERR_HIP, "iteration loop ended without a final state");
    if (status == OFX_OK) {
        int n_of[OFX_MAX_GROUP], k0_of[OFX_MAX_GROUP], u_of[OFX_MAX_GROUP];
        bool any = false;
        for (int g = 0; g < G; g++) {
            const bool inside = fin[g].n > 0 && fin[g].n - fin[g].k0 < fin[g].ucnt;
            n_of[g] = inside ? fin[g].n : -1;
            k0_of[g] = fin[g].k0;
            u_of[g] = fin[g].unit;
            any = any || inside;
            n_out[g] = fin[g].n;
            err_out[g] = fin[g].error;
            unit_out[g] = fin[g].n > 0 ? fin[g].unit : -1;
            if (inside_out) inside_out[g] = inside ? 1 : 0;
        }
        if (any) status = redo(n_of, k0_of, u_of);
    }
    if (status != OFX_OK) {
        (void) hipStreamSynchronize(ctx->stream);       // nothing of this loop in flight when the caller resets the arena
        return status;
    }
    if (ms_out) {
        OFX_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
        OFX_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
        OFX_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev_t0, ctx->ev_t1));
    }
    return OFX_OK;
}
