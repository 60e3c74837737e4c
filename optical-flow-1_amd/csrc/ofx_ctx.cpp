// ofx_ctx.cpp -- context, device arena, options, error reporting (host side of libofx.so).
#include "ofx_internal.h"

#include <cstdarg>
#include <cstdlib>
#include <new>

static const size_t kSlabAlign = 256;
static const size_t kMinSlab = 32u << 20;
static const size_t kMaxSlabs = 24;

int ofx_fail(ofx_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->errmsg, sizeof(ctx->errmsg), fmt, ap);
        va_end(ap);
    }
    return status;
}

extern "C" const char *ofx_strerror(int status)
{
    switch (status) {
    case OFX_OK:        return "ok";
    case OFX_ERR_ARG:   return "invalid argument";
    case OFX_ERR_SIGMA: return "GaussianSmooth: sigma too large";
    case OFX_ERR_NOMEM: return "out of memory";
    case OFX_ERR_HIP:   return "HIP runtime error";
    case OFX_ERR_NODEV: return "no usable gfx950 device";
    default:            return "unknown status";
    }
}

extern "C" const char *ofx_last_error(const ofx_ctx *ctx) { return ctx ? ctx->errmsg : ""; }

extern "C" int ofx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void ofx_ctx_destroy(ofx_ctx *ctx);

extern "C" int ofx_ctx_create(ofx_ctx **out, int device, int precision)
{
    if (!out) return OFX_ERR_ARG;
    *out = nullptr;
    if (precision != OFX_F64 && precision != OFX_F32) return OFX_ERR_ARG;
    // OFX_TRACE_INIT=1: where the time of the first context of a process goes (stderr), for the front-ends' end-to-end budget
    const bool trace = getenv("OFX_TRACE_INIT") != nullptr;
    double t_prev = ofx_now_ms();
    auto mark = [&](const char *what) {
        if (!trace) return;
        const double t = ofx_now_ms();
        fprintf(stderr, "ofx_ctx_create: %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return OFX_ERR_NODEV;
    mark("hipGetDeviceCount (runtime)");
    if (hipSetDevice(device) != hipSuccess) return OFX_ERR_NODEV;
    mark("hipSetDevice");

    ofx_ctx *ctx = new (std::nothrow) ofx_ctx();
    if (!ctx) return OFX_ERR_NOMEM;
    ctx->device = device;
    ctx->precision = precision;
    ctx->cur_slab = 0;
    ctx->cur_used = 0;
    ctx->call_bytes = 0;
    ctx->profile = 0;
    ctx->rows_per_wave = 0;     // 0 = pick per level
    ctx->rows_per_wave2 = 0;
    ctx->fuse2 = 1;             // two TV-L1 iterations per launch
    ctx->store_a = 1;
    ctx->concurrency = 1;
    ctx->lockstep = 0;
    ctx->warp_lds = 1;
    ctx->gauss_fused = 1;
    ctx->relaxed_dual = 0;
    ctx->tile = 0;              // measured: no faster than the marching strips on the small levels, off by default
    ctx->tile_max_px = 0;
    ctx->nt_stores = 0;
    ctx->chunk = 0;             // 0 = pick per level
    ctx->spin_us = 150;
    ctx->rows_slots = 0;
    ctx->rof_pipe = 1;
    ctx->rof_window = 0;
    ctx->chi_fuse = 1;
    ctx->fuse3 = 2;
    ctx->fuse3_min_px = 0.0;
    ctx->fuse3_cursor = 1;
    ctx->fuse3_afac1 = 0;
    ctx->fuse3_afac2 = 0;
    ctx->rows_per_wave3 = 0;
    ctx->rows3_max = 32;
    ctx->fixed_work = 0;
    ctx->sor_exact = 1;
    ctx->sor_batch = 0;
    ctx->sor_fuse = 0;
    ctx->sor_tile = 0;
    ctx->sor_tile_w = 0;
    ctx->sor_wave_levels = 1;
    ctx->sor_wave_p = 0;
    ctx->sor_window = 0;
    ctx->sor_rows = 0;
    ctx->sor_spw = 0;
    ctx->sor_lds = 1;
    ctx->mem_budget = 0;
    ctx->poll_seq = 0;
    ctx->errmsg[0] = 0;
    memset(&ctx->stats, 0, sizeof(ctx->stats));

    ctx->stream = nullptr;
    ctx->d_err = nullptr;
    ctx->d_state = nullptr;
    ctx->h_state = nullptr;
    ctx->h_aux = nullptr;
    ctx->ev_t0 = ctx->ev_t1 = nullptr;
    for (int i = 0; i < OFX_NPOLL; i++) ctx->ev_poll[i] = nullptr;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    mark("hipStreamCreate");
    static_assert(sizeof(OfxLoopDev) <= OFX_STATE_BYTES && OFX_STATE_BYTES % 512 == 0, "state block too small");
    static_assert(OFX_ULOG >= OFX_TVL1_MAX_ITERATIONS + 16, "unit log too short");
    ok = ok && hipMalloc((void **) &ctx->d_state, OFX_STATE_BYTES + sizeof(double) * OFX_TVL1_MAX_ITERATIONS * OFX_NSHARD) == hipSuccess;
    if (ok) ctx->d_err = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->d_state) + OFX_STATE_BYTES);
    ctx->d_err_cap = OFX_TVL1_MAX_ITERATIONS;
    mark("hipMalloc (loop state)");
    // the poll ring, and behind it OFX_MAX_GROUP doubles of pinned scratch (h_aux: the occlusion solver's per-triple errors); the
    // ring starts zeroed -- ofx_loop_wait_poll reads a record's `seq` == 0 as "not published yet"
    ok = ok && hipHostMalloc((void **) &ctx->h_state, sizeof(OfxIterState) * OFX_NPOLL * OFX_MAX_GROUP + sizeof(double) * OFX_MAX_GROUP,
                             hipHostMallocDefault) == hipSuccess;
    if (ok) {
        memset(ctx->h_state, 0, sizeof(OfxIterState) * OFX_NPOLL * OFX_MAX_GROUP + sizeof(double) * OFX_MAX_GROUP);
        ctx->h_aux = reinterpret_cast<double *>(ctx->h_state + OFX_NPOLL * OFX_MAX_GROUP);
    }
    mark("hipHostMalloc (poll ring)");
    for (int i = 0; ok && i < OFX_NPOLL; i++)
        ok = hipEventCreateWithFlags(&ctx->ev_poll[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->ev_t0) == hipSuccess && hipEventCreate(&ctx->ev_t1) == hipSuccess;
    mark("hipEventCreate x 6");
    if (!ok) {
        (void) hipGetLastError();
        ofx_ctx_destroy(ctx);        // every handle is either valid or still null
        return OFX_ERR_HIP;
    }
    *out = ctx;
    return OFX_OK;
}

extern "C" void ofx_ctx_destroy(ofx_ctx *ctx)
{
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->slabs) (void) hipFree(s.base);
    if (ctx->d_state) (void) hipFree(ctx->d_state);            // d_err lives in the same allocation
    if (ctx->h_state) (void) hipHostFree(ctx->h_state);
    for (int i = 0; i < OFX_NPOLL; i++)
        if (ctx->ev_poll[i]) (void) hipEventDestroy(ctx->ev_poll[i]);
    if (ctx->ev_t0) (void) hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void) hipEventDestroy(ctx->ev_t1);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    (void) hipGetLastError();
    delete ctx;
}

extern "C" void *ofx_ctx_stream(const ofx_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }
extern "C" int ofx_ctx_precision(const ofx_ctx *ctx) { return ctx ? ctx->precision : -1; }

extern "C" int ofx_ctx_synchronize(ofx_ctx *ctx)
{
    if (!ctx) return OFX_ERR_ARG;
    OFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OFX_OK;
}

extern "C" int ofx_set_option(ofx_ctx *ctx, const char *name, double value)
{
    if (!ctx || !name) return OFX_ERR_ARG;
    if (!strcmp(name, "profile")) { ctx->profile = value != 0; return OFX_OK; }
    if (!strcmp(name, "fixed_work")) { ctx->fixed_work = value != 0; return OFX_OK; }
    if (!strcmp(name, "rows_per_wave")) {
        if (value < 0 || value > 4096) return ofx_fail(ctx, OFX_ERR_ARG, "rows_per_wave out of range");
        ctx->rows_per_wave = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "rows_per_wave2")) {
        if (value < 0 || value > 4096) return ofx_fail(ctx, OFX_ERR_ARG, "rows_per_wave2 out of range");
        ctx->rows_per_wave2 = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "concurrency")) {
        if (value < 1 || value > 64) return ofx_fail(ctx, OFX_ERR_ARG, "concurrency out of range");
        ctx->concurrency = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "tile")) {
        if (value != 0 && value != 4 && value != 6) return ofx_fail(ctx, OFX_ERR_ARG, "tile must be 0, 4 or 6");
        ctx->tile = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "tile_max_px")) {
        if (value < 0) return ofx_fail(ctx, OFX_ERR_ARG, "tile_max_px out of range");
        ctx->tile_max_px = value;
        return OFX_OK;
    }
    if (!strcmp(name, "relaxed_dual")) { ctx->relaxed_dual = value != 0; return OFX_OK; }
    if (!strcmp(name, "gauss_fused")) { ctx->gauss_fused = (value == 2 || value == 3) ? (int) value : (value != 0); return OFX_OK; }   // 2: the generic-radius fused kernel; 3: radius-templated, zoom_out not fused
    if (!strcmp(name, "warp_lds")) { ctx->warp_lds = value != 0; return OFX_OK; }
    if (!strcmp(name, "nt_stores")) { ctx->nt_stores = (int) value; return (value >= 0 && value <= 2) ? OFX_OK : ofx_fail(ctx, OFX_ERR_ARG, "nt_stores = 0 | 1 | 2"); }
    if (!strcmp(name, "lockstep")) {
        if (value < 0 || value > OFX_MAX_GROUP) return ofx_fail(ctx, OFX_ERR_ARG, "lockstep out of range");
        ctx->lockstep = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_exact")) {
        if (value != 0 && value != 1 && value != 2) return ofx_fail(ctx, OFX_ERR_ARG, "sor_exact must be 0, 1 or 2");
        ctx->sor_exact = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_fuse")) {
        if ((value < -1 || value > 4) && value != 9) return ofx_fail(ctx, OFX_ERR_ARG, "sor_fuse must be -1 .. 4 (or 9)");
        ctx->sor_fuse = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_tile")) {
        if (value < 0 || value > 3) return ofx_fail(ctx, OFX_ERR_ARG, "sor_tile must be 0 .. 3");
        ctx->sor_tile = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_tile_w")) {
        if (value < 0 || value > 65536) return ofx_fail(ctx, OFX_ERR_ARG, "sor_tile_w out of range");
        ctx->sor_tile_w = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_wave_p")) {
        if (value < 0 || value > 8) return ofx_fail(ctx, OFX_ERR_ARG, "sor_wave_p out of range");
        ctx->sor_wave_p = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_wave_levels")) {
        if (value < 0 || value > 64) return ofx_fail(ctx, OFX_ERR_ARG, "sor_wave_levels out of range");
        ctx->sor_wave_levels = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_rows")) {
        if (value < 0 || value > 4096) return ofx_fail(ctx, OFX_ERR_ARG, "sor_rows out of range");
        ctx->sor_rows = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_lds")) {
        if (value != 0 && value != 1 && value != 2) return ofx_fail(ctx, OFX_ERR_ARG, "sor_lds must be 0, 1 or 2");
        ctx->sor_lds = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_spw")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return ofx_fail(ctx, OFX_ERR_ARG, "sor_spw must be 0, 1, 2 or 4");
        ctx->sor_spw = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_window")) {
        if (value < 0 || value > 4096) return ofx_fail(ctx, OFX_ERR_ARG, "sor_window out of range");
        ctx->sor_window = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "sor_batch")) {
        if (value < 0 || value > 4096) return ofx_fail(ctx, OFX_ERR_ARG, "sor_batch out of range");
        ctx->sor_batch = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "mem_budget")) {
        if (value < 0) return ofx_fail(ctx, OFX_ERR_ARG, "mem_budget out of range");
        ctx->mem_budget = value;
        return OFX_OK;
    }
    if (!strcmp(name, "fuse2")) { ctx->fuse2 = value != 0; return OFX_OK; }
    if (!strcmp(name, "store_a")) {
        if (value != 0 && value != 1 && value != 2) return ofx_fail(ctx, OFX_ERR_ARG, "store_a must be 0, 1 or 2");
        ctx->store_a = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "rows_slots")) { ctx->rows_slots = (int) value; return OFX_OK; }
    if (!strcmp(name, "rof_pipe")) { ctx->rof_pipe = value != 0; return OFX_OK; }
    if (!strcmp(name, "rof_window")) {
        if (value != 0 && value != 24 && value != 10) return ofx_fail(ctx, OFX_ERR_ARG, "rof_window must be 0, 10 or 24");
        ctx->rof_window = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "chi_fuse")) { ctx->chi_fuse = value != 0; return OFX_OK; }
    if (!strcmp(name, "fuse3")) {
        if (value != 0 && value != 1 && value != 2) return ofx_fail(ctx, OFX_ERR_ARG, "fuse3 must be 0, 1 or 2");
        ctx->fuse3 = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "fuse3_min_px")) { ctx->fuse3_min_px = value; return OFX_OK; }
    if (!strcmp(name, "fuse3_cursor")) { ctx->fuse3_cursor = value != 0; return OFX_OK; }
    if (!strcmp(name, "fuse3_afac1")) { ctx->fuse3_afac1 = value; return value >= 0 ? OFX_OK : ofx_fail(ctx, OFX_ERR_ARG, "fuse3_afac1 < 0"); }
    if (!strcmp(name, "fuse3_afac2")) { ctx->fuse3_afac2 = value; return value >= 0 ? OFX_OK : ofx_fail(ctx, OFX_ERR_ARG, "fuse3_afac2 < 0"); }
    if (!strcmp(name, "rows_per_wave3")) { ctx->rows_per_wave3 = (int) value; return OFX_OK; }
    if (!strcmp(name, "rows3_max")) { ctx->rows3_max = (int) value; return OFX_OK; }
    if (!strcmp(name, "spin_us")) {
        if (value < 0 || value > 1e6) return ofx_fail(ctx, OFX_ERR_ARG, "spin_us out of range");
        ctx->spin_us = (int) value;
        return OFX_OK;
    }
    if (!strcmp(name, "chunk")) {
        if (value < 0 || value > OFX_TVL1_MAX_ITERATIONS) return ofx_fail(ctx, OFX_ERR_ARG, "chunk out of range");
        ctx->chunk = (int) value;
        return OFX_OK;
    }
    return ofx_fail(ctx, OFX_ERR_ARG, "unknown option '%s'", name);
}

extern "C" int ofx_get_stats(const ofx_ctx *ctx, ofx_stats *out)
{
    if (!ctx || !out) return OFX_ERR_ARG;
    *out = ctx->stats;
    return OFX_OK;
}

// ---- arena ---------------------------------------------------------------------------------------
// Per-call bump allocation out of a few big hipMalloc slabs.  A call that outgrows the slabs gets an
// additional one (at least as large as everything the call has allocated so far, so their number grows
// with the logarithm of the workspace).  The same request sequence walks the same slabs the same way, so in
// steady state (same image size call after call) no hipMalloc/hipFree happens at all -- the slabs are
// deliberately NOT merged after the first call: freeing and re-allocating many GB costs seconds (measured:
// 2.2 s for 4 contexts x 4.5 GB) and would land in the caller's second call.  Only a context that has
// collected many slabs from calls of changing sizes is coalesced.
void ofx_arena_reset(ofx_ctx *ctx)
{
    if (ctx->slabs.size() > kMaxSlabs) {
        size_t total = 0;
        for (auto &s : ctx->slabs) { total += s.bytes; (void) hipFree(s.base); }
        ctx->slabs.clear();
        char *p = nullptr;
        if (hipMalloc((void **) &p, total) == hipSuccess) ctx->slabs.push_back({p, total});
        else (void) hipGetLastError();
    }
    ctx->cur_slab = 0;
    ctx->cur_used = 0;
    ctx->call_bytes = 0;
}

int ofx_arena_alloc(ofx_ctx *ctx, size_t bytes, void **out)
{
    static const long skew = getenv("OFX_ARENA_SKEW") ? atol(getenv("OFX_ARENA_SKEW")) : 0;   // A/B knob only
    bytes += (size_t) skew;
    bytes = (bytes + kSlabAlign - 1) / kSlabAlign * kSlabAlign;
    if (bytes == 0) bytes = kSlabAlign;
    while (ctx->cur_slab < ctx->slabs.size()) {
        OfxSlab &s = ctx->slabs[ctx->cur_slab];
        if (ctx->cur_used + bytes <= s.bytes) {
            *out = s.base + ctx->cur_used;
            ctx->cur_used += bytes;
            ctx->call_bytes += bytes;
            return OFX_OK;
        }
        ctx->cur_slab++;
        ctx->cur_used = 0;
    }
    size_t want = bytes > kMinSlab ? bytes : kMinSlab;
    if (want < ctx->call_bytes) want = ctx->call_bytes;        // geometric-ish growth inside one call
    char *p = nullptr;
    hipError_t e = hipMalloc((void **) &p, want);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        return ofx_fail(ctx, OFX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    ctx->slabs.push_back({p, want});
    ctx->cur_slab = ctx->slabs.size() - 1;
    ctx->cur_used = bytes;
    ctx->call_bytes += bytes;
    *out = p;
    return OFX_OK;
}
