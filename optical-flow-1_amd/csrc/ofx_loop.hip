// ofx_loop.hip -- finalize kernel and host helpers of the convergence-loop machinery (ofx_loop.h).
#include "ofx_loop.h"

// One block (16 waves) per problem of a lockstep group.  Scans the error slots of sweeps [start, launched)
// and publishes how many sweeps really ran (the reference's n), whether the loop is over, and the criterion
// value at exit.
__global__ __launch_bounds__(1024) void k_loop_finalize(const double *__restrict__ err, int slots_per_problem, int start,
                                                        int launched, int max_iter, int size, double thr, int crit,
                                                        double afac, OfxIterState *st, OfxIterState *host_st, int seq)
{
    extern __shared__ double s_err[];
    err += (size_t) blockIdx.x * slots_per_problem * OFX_NSHARD;
    st += blockIdx.x;
    host_st += blockIdx.x;
    if (st->done) {                                         // an earlier chunk already ended the loop
        if (threadIdx.x == 0) {
            host_st->n = st->n;
            host_st->error = st->error;
            host_st->apred = st->apred;
            host_st->done = st->done;
            __threadfence_system();                         // the record before its sequence number (the host may be spinning on it)
            __hip_atomic_store(&host_st->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = start + w; k < launched; k += 16) {
        const double e = loop_error_from_sum(wave_allreduce_sum(err[(size_t) k * OFX_NSHARD + lane]), size, crit);
        if (lane == 0) s_err[k - start] = e;
    }
    __syncthreads();
    if (w == 0) {                                           // every lane of wave 0 computes the same record
        int n = launched, done = (launched >= max_iter);
        double error = launched > start ? s_err[launched - start - 1] : st->error;
        for (int k = start; k < launched; k++) {
            if (!(s_err[k - start] > thr)) { n = k + 1; done = 1; error = s_err[k - start]; break; }
        }
        // odd n: iteration n - 1 was the first of a fused pair; its launch stored the intermediate state iff the
        // error of iteration n - 2 was within afac of the threshold (tvl1_store_a)
        int apred = 0;
        if (afac > 0.0 && done && (n & 1) && n >= 3) {
            const double e = loop_error_from_sum(wave_allreduce_sum(err[(size_t) (n - 2) * OFX_NSHARD + lane]), size, crit);
            apred = e <= thr * afac;
        }
        if (lane == 0) {
            st->n = n;
            st->done = done;
            st->error = error;
            st->apred = apred;
            host_st->n = n;                                 // pinned host memory: visible once the kernel retires ...
            host_st->error = error;
            host_st->apred = apred;
            host_st->done = done;
            __threadfence_system();                         // ... and, for a host that spins on `seq`, as soon as seq is
            __hip_atomic_store(&host_st->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The same for a CURSOR loop (OfxLoopDev): the range of iterations a chunk of launches ran is device state -- from scanned[g]
// to the cursor the chunk's last launch (index launch_end - 1) left behind.  When the loop is over the record also names the
// launch unit that contains the stopping iteration (the largest j with ulog[j] <= n - 1: later launches were no-ops and logged
// the unchanged cursor), so that the host can re-run its first iterations from its input.
__global__ __launch_bounds__(1024) void k_loop_finalize_cursor(const double *__restrict__ err, int slots_per_problem, int launch_end,
                                                               int max_iter, int size, double thr, int crit, OfxLoopDev *dev,
                                                               OfxIterState *host_st, int seq)
{
    extern __shared__ double s_err[];
    const int g = blockIdx.x;
    err += (size_t) g * slots_per_problem * OFX_NSHARD;
    OfxIterState *st = dev->st + g;
    host_st += g;
    if (st->done) {
        if (threadIdx.x == 0) {
            host_st->n = st->n;
            host_st->error = st->error;
            host_st->apred = 0;
            host_st->unit = st->unit;
            host_st->k0 = st->k0;
            host_st->ucnt = st->ucnt;
            host_st->done = st->done;
            __threadfence_system();
            __hip_atomic_store(&host_st->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int start = dev->scanned[g], launched = dev->cursor[launch_end & 1][g];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = start + w; k < launched; k += 16) {
        const double e = loop_error_from_sum(wave_allreduce_sum(err[(size_t) k * OFX_NSHARD + lane]), size, crit);
        if (lane == 0) s_err[k - start] = e;
    }
    __syncthreads();
    if (w == 0) {
        int n = launched, done = (launched >= max_iter);
        double error = launched > start ? s_err[launched - start - 1] : st->error;
        for (int k = start; k < launched; k++) {
            if (!(s_err[k - start] > thr)) { n = k + 1; done = 1; error = s_err[k - start]; break; }
        }
        int unit = 0;
        if (done && n > 0)
            for (int j = launch_end - 1; j >= 0; j--)
                if (dev->ulog[g][j] <= n - 1) { unit = j; break; }
        if (lane == 0) {
            const int k0 = dev->ulog[g][unit];
            const int next = unit + 1 < launch_end ? dev->ulog[g][unit + 1] : launched;
            st->n = n;
            st->done = done;
            st->error = error;
            st->unit = unit;
            st->k0 = k0;
            st->ucnt = next - k0;
            dev->scanned[g] = launched;
            host_st->n = n;
            host_st->error = error;
            host_st->apred = 0;
            host_st->unit = unit;
            host_st->k0 = k0;
            host_st->ucnt = next - k0;
            host_st->done = done;
            __threadfence_system();
            __hip_atomic_store(&host_st->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int ofx_loop_finalize_cursor(ofx_ctx *ctx, const LoopSpec &L, int G, int slots_per_problem, int launch_end, int max_range,
                             OfxIterState *host_slot, int seq)
{
    const size_t shmem = sizeof(double) * (size_t) max_range;
    hipLaunchKernelGGL(k_loop_finalize_cursor, dim3(G), dim3(1024), shmem, ctx->stream, (const double *) ctx->d_err, slots_per_problem,
                       launch_end, L.max_iter, L.size, L.thr, L.crit, reinterpret_cast<OfxLoopDev *>(ctx->d_state), host_slot, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}

int ofx_loop_finalize_group(ofx_ctx *ctx, const LoopSpec &L, int G, int slots_per_problem, int start, int launched,
                            OfxIterState *host_slot, int seq)
{
    const size_t shmem = sizeof(double) * (size_t) (launched - start);
    hipLaunchKernelGGL(k_loop_finalize, dim3(G), dim3(1024), shmem, ctx->stream, (const double *) ctx->d_err,
                       slots_per_problem, start, launched, L.max_iter, L.size, L.thr, L.crit, L.afac, ctx->d_state, host_slot, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofx_fail(ctx, OFX_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
    return OFX_OK;
}

int ofx_loop_reserve(ofx_ctx *ctx, int max_iter)
{
    if (max_iter <= ctx->d_err_cap) return OFX_OK;
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_state) (void) hipFree(ctx->d_state);
    ctx->d_state = nullptr;
    ctx->d_err = nullptr;
    ctx->d_err_cap = 0;
    OFX_HIP(ctx, hipMalloc((void **) &ctx->d_state, OFX_STATE_BYTES + sizeof(double) * (size_t) max_iter * OFX_NSHARD));
    ctx->d_err = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->d_state) + OFX_STATE_BYTES);
    ctx->d_err_cap = max_iter;
    return OFX_OK;
}

// zero the loop state and the first `slots` error slots: one memset (the two arrays share an allocation)
int ofx_loop_clear(ofx_ctx *ctx, size_t slots)
{
    OFX_HIP(ctx, hipMemsetAsync(ctx->d_state, 0, OFX_STATE_BYTES + sizeof(double) * slots * OFX_NSHARD, ctx->stream));
    return OFX_OK;
}

// Wait for the poll record of `slot` (G problems) that finalize launch number `seq` publishes.  The host first spins on the
// records' sequence numbers in pinned memory -- the finalize kernel writes them last, behind a system-scope fence, so a
// record whose number is there is complete -- for at most ctx->spin_us microseconds: that sees the result ~2 us after
// the kernel wrote it, where hipEventSynchronize needs 20-40 us to wake the thread.  A chunk that takes longer than that is
// long enough for the wake-up not to matter: fall back to the event.
int ofx_loop_wait_poll(ofx_ctx *ctx, int slot, int G, int seq)
{
    if (ctx->spin_us > 0) {
        const double t_end = ofx_now_ms() + ctx->spin_us * 1e-3;
        volatile OfxIterState *rec = ctx->h_state + (size_t) slot * OFX_MAX_GROUP;
        for (int spins = 0;; spins++) {
            bool all = true;
            for (int g = 0; g < G && all; g++) all = __atomic_load_n((const int *) &rec[g].seq, __ATOMIC_ACQUIRE) == seq;
            if (all) return OFX_OK;
            if ((spins & 63) == 63 && ofx_now_ms() > t_end) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            __asm__ __volatile__("" ::: "memory");
#endif
        }
    }
    OFX_HIP(ctx, hipEventSynchronize(ctx->ev_poll[slot]));
    return OFX_OK;
}
