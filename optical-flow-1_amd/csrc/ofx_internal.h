// ofx_internal.h -- private to libofx.so (host side of the HIP path; nothing here is ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>

#include "ofx.h"

#define OFX_NSHARD 64          // error-accumulator shards per iteration slot (= wave size)
#define OFX_NPOLL  4           // in-flight convergence polls
#define OFX_MAX_GROUP 16       // image pairs one context can solve in lockstep (TV-L1 groups)

// What the finalize kernel publishes to the host after every chunk of iterations.
struct OfxIterState {
    int    n;        // iterations that did real work so far (== reference's `n`)
    int    done;     // 1: the stopping test fired (or MAX reached)
    double error;    // value of the stopping criterion after iteration n
    int    apred;    // TV-L1 fused pairs: n is odd and the launch that ran iteration n - 1 had predicted the stop and
                     // stored its intermediate state (tvl1_store_a in ofx_tvl1.hip, same formula)
    int    seq;      // host copy only: written LAST (after a system-scope fence) with the poll's sequence number, so the host
                     // can spin on it instead of sleeping in hipEventSynchronize (0 = not published yet)
    // cursor loops (ofx_run_loop_cursor): the launch unit that contains iteration n - 1 -- its launch index, first iteration
    // and iteration count; meaningful when done
    int    unit, k0, ucnt;
};

// Device-resident part of a cursor loop (TV-L1, three-iteration units that shrink near the end of a loop): the launch units of
// a pair have no fixed size, so where a launch starts is device state.  Launch L of pair g reads cursor[L & 1][g], decides its
// iteration count from the previous iteration's error, and ONE thread of it writes cursor[(L + 1) & 1][g] and ulog[g][L] (the
// other blocks of launch L may not have read their cursor yet: two copies by launch parity).  scanned[g]: iterations the
// finalize kernels have looked at.  The whole block is zeroed with the error slots at the start of every loop.
#define OFX_ULOG 320           // launches of one loop: a unit may be a single iteration, so up to OFX_TVL1_MAX_ITERATIONS of them (+ no-ops behind the stop)
struct OfxLoopDev {
    OfxIterState st[OFX_MAX_GROUP];
    int cursor[2][OFX_MAX_GROUP];
    int scanned[OFX_MAX_GROUP];
    int ulog[OFX_MAX_GROUP][OFX_ULOG];
};

#define OFX_STATE_BYTES 22016  // >= sizeof(OfxLoopDev), a multiple of 512: keeps the error slots 512-byte aligned

struct OfxSlab {
    char  *base;
    size_t bytes;
};

struct ofx_ctx {
    int device;
    int precision;
    hipStream_t stream;

    // bump arena for all per-call device arrays (reset at the start of each API call)
    std::vector<OfxSlab> slabs;
    size_t cur_slab;
    size_t cur_used;
    size_t call_bytes;      // bytes handed out during this call (to coalesce next time)

    // convergence machinery
    // one allocation, [state: OFX_STATE_BYTES][error slots], so that a loop clears both with ONE memset (ofx_loop_clear)
    double       *d_err;    // [d_err_cap][OFX_NSHARD] per-iteration squared-update sums (= d_state + OFX_STATE_BYTES)
    int           d_err_cap;
    OfxIterState *d_state;  // device copy [OFX_MAX_GROUP], start of the allocation
    OfxIterState *h_state;  // pinned ring [OFX_NPOLL][OFX_MAX_GROUP]
    double       *h_aux;    // pinned, OFX_MAX_GROUP doubles behind the ring (same allocation): results that are not loop records
    hipEvent_t    ev_poll[OFX_NPOLL];
    hipEvent_t    ev_t0, ev_t1;

    // options
    int profile;
    int rows_per_wave;
    int rows_per_wave2;
    int fuse2;
    int store_a;        // TV-L1 fused pairs: 1 (default) store the intermediate state when a stop is plausible, 0 never
                        // (odd stops are recomputed), 2 always (tests)
    int concurrency;    // contexts expected to share the device (tuning hint, default 1)
    int lockstep;       // pairs per lockstep group in ofx_tvl1_batch_dev (0 = default)
    int nt_stores;      // fused TV-L1 kernel: 0 (default) non-temporal stores once a launch's working set exceeds the Infinity Cache, 1 always, 2 never
    int relaxed_dual;   // TV-L1 in double storage: 1 = the fast mode's dual stage (sqrt(x^2 + y^2), one reciprocal per denominator) --
                        // the "tolerance" mode, AEPE vs the reference ~1e-12 px on the BASELINE configs (profiles/r03_b_relaxed_dual_accuracy.jsonl),
                        // not bit-identical; 0 (default) strict
    int tile;           // TV-L1 small levels: K iterations per launch on 2-D tiles (k_tvl1_tile): 4 | 6 | 0 = off (default)
    double tile_max_px; // ... for levels of at most this many pixels x pairs (0 = default 200 000)
    int gauss_fused;    // pyramids of lockstep groups: row + column pass of the Gaussian in one launch through LDS (1 default)
    int warp_lds;       // 1 (default): TV-L1 warp with the taps staged through LDS; 0: gathered from global memory
    int chunk;
    int fuse3;          // TV-L1: three iterations per launch (k_tvl1_iter3): 0 never, 1 levels of >= fuse3_min_px pixels x pairs, 2 auto (default)
    double fuse3_min_px;
    int fuse3_cursor;   // 1 (default): k_tvl1_iter3 as a cursor loop -- units shrink to 2 / 1 iterations near the end of a loop; 0: fixed units of 3
    double fuse3_afac1, fuse3_afac2;   // ... error / threshold ratios below which a unit runs 1 / 2 iterations (0 = 1.2 / 1.5)
    int rows_per_wave3; // strip height of k_tvl1_iter3 (0 = tvl1_pick_rows3)
    int rows3_max;      // tallest strip tvl1_pick_rows3 considers when contexts share the device
    int chi_fuse;       // Solver_wrt_chi: 1 = CHI_N iterations per launch on LDS tiles (default), 0 = two launches per iteration
    int rof_window;     // steps per launch of the ROF box sweeps: 10 (default, also 0) | 24
    int rof_pipe;       // ROF box sweeps (ofx_occ.hip): 1 = all iterations of a call in flight (default), 0 = one at a time
    int rows_slots;     // strip-height model of the fused kernel when contexts share the device: waves per "round" (0 = 1024)
    int spin_us;        // convergence polls: microseconds the host spins on the pinned record before it falls back to
                        // hipEventSynchronize (default 150; 0 = never spin)
    int fixed_work;
    int sor_exact;      // 1: reference sweep order, windowed launches; 2: same, one launch per time step; 0: colour order
    int sor_fuse;       // sor_exact = 0: K sweeps per launch on LDS tiles (ofx_sor_tile.hip): 0 = automatic (default), 1..4 = K,
                        // -1 = one launch per colour and sweep (k_hs_sor / k_brox_sor; single pairs only)
    int sor_tile;       // tile geometry of the sor_exact = 0 sweeps (0 = default; ofx_sor_tile.hip)
    int sor_tile_w;     // Brox, sor_exact = 0: columns per tile of the checkerboard sweeps (0 = 128, the maximum)
    int sor_wave_p;     // ... anti-diagonals of operand prefetch in k_brox_wave (0 = 4)
    int sor_wave_levels;// Brox, sor_exact = 0: pyramid levels 0 .. n - 1 use the checkerboard-of-tiles sweeps, the coarser ones red-black (default 1)
    int sor_batch;      // sweeps in flight per batch in exact mode (0 = default)
    int sor_window;     // time steps per launch of the windowed exact mode (0 = 8)
    int sor_rows;       // rows per block (workgroup) of a sweep in the windowed exact mode (0 = 64)
    int sor_lds;        // windowed exact sweeps: 0 = one global round trip per step, 2 = the launch window staged in LDS, 1 (default) = by
                        // measurement: LDS for lone Horn-Schunck solves, global otherwise (sor_use_lds)
    int sor_spw;        // sweeps per workgroup of the windowed exact kernels (0 = automatic: 1 alone, 2 in lockstep groups)
    double mem_budget;  // bytes all contexts of a batch may use for level arrays (0 = half of the free device memory)
    unsigned long long poll_seq;

    ofx_stats stats;
    char errmsg[256];
};

// ---- error plumbing ------------------------------------------------------------------------
int ofx_fail(ofx_ctx *ctx, int status, const char *fmt, ...);

#define OFX_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return ofx_fail((ctx), e__ == hipErrorOutOfMemory ? OFX_ERR_NOMEM : OFX_ERR_HIP,     \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,   \
                            __LINE__);                                                           \
    } while (0)

#define OFX_TRY(expr)                                                                            \
    do {                                                                                         \
        int s__ = (expr);                                                                        \
        if (s__ != OFX_OK) return s__;                                                           \
    } while (0)

// ---- arena -----------------------------------------------------------------------------------
void ofx_arena_reset(ofx_ctx *ctx);
int  ofx_arena_alloc(ofx_ctx *ctx, size_t bytes, void **out);

template <typename T>
static inline int ofx_alloc(ofx_ctx *ctx, size_t count, T **out)
{
    void *p = nullptr;
    int s = ofx_arena_alloc(ctx, count * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return s;
}

// ---- storage-type traits -----------------------------------------------------------------------
template <typename T> struct Pix;
template <> struct Pix<double> { using v2 = double2; using v4 = double4; };
template <> struct Pix<float>  { using v2 = float2;  using v4 = float4;  };

static inline int ofx_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- API entry boilerplate ---------------------------------------------------------------------
#define OFX_ENTER(ctx)                                                                           \
    do {                                                                                         \
        if (!(ctx)) return OFX_ERR_ARG;                                                          \
        if (hipSetDevice((ctx)->device) != hipSuccess)                                           \
            return ofx_fail((ctx), OFX_ERR_NODEV, "hipSetDevice(%d) failed", (ctx)->device);     \
        (ctx)->errmsg[0] = 0;                                                                    \
        ofx_arena_reset(ctx);                                                                    \
    } while (0)

static inline double ofx_now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
