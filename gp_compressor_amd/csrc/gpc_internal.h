// gpc_internal.h -- shared host-side definitions of libgpc_hip.so (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>

#include "../../include/gpc.h"

// A Cholesky pivot <= GPC_PIVOT_RTOL * (sigma_f^2 + noise) is reported as GPC_STATUS_NOT_SPD.  Eigen::LLT tests
// `pivot <= 0` (and the reference never looks at the result, /root/reference/src/gaussian_process.cpp:22); with a
// kernel + noise >= 0 Gram matrix a non-positive pivot can only come from cancellation, where the sign of the
// 1e-19 residue is an accident of the summation order.  A relative threshold makes the report deterministic.
#define GPC_PIVOT_RTOL 1e-14

struct gpc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;  // created with the context
    hipStream_t stream = nullptr;      // stream in use (own_stream or the caller's)
    int num_cus = 256;
    // grow-only device workspace (K / L factors of the generic dense kernel, variance scratch, grid tables)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_len = 0;                 // ... and its length (0: to the end of the workspace); GPC_POISON_LDS poisons only this region
    size_t ws_off = 0;                 // offset of the region the launch in hand may use (the host-pointer pipeline runs the kernels of
                                       // consecutive chunks on two streams, each in its own half; 0 everywhere else)
    int32_t* tickets = nullptr;        // 64 counters (allocated on first use): patches handed out one at a time where their cost varies (dense_mfma_big.hip)
    // host-pointer entries (gpc_api.hip, dense_host): a grow-only device arena for the batch, pinned staging buffers for
    // pageable caller memory, and two copy streams so that the upload of chunk c+1 and the download of chunk c-1 run on the
    // SDMA engines while the kernel works on chunk c
    void* io = nullptr;
    size_t io_bytes = 0;
    void* pin_in = nullptr;
    size_t pin_in_bytes = 0;
    void* pin_out = nullptr;
    size_t pin_out_bytes = 0;
    hipStream_t s_in = nullptr, s_out = nullptr, s_c2 = nullptr;   // copy-in, copy-out, second compute stream
    bool pipe_active = false;          // a two-stream host-pointer call is in flight on own_stream / s_c2 (under mu): see gpc_ws_reserve
    unsigned foreign_gen = 0;          // ... and the number of calls of other threads that touched the workspace meanwhile
    hipEvent_t ev[3][16] = {};       // [.][0 .. 7] the chunks of the host-pointer pipeline, [.][10] the pipeline against other threads' calls, [.][13] the class fork, [0][15] the arena
    std::mutex host_mu;                // one host-pointer call at a time per context (they share the arena)
    // size classes of the last batch gpc_project_cloud produced on this context (its `off` buffer, how many of its P patches have
    // <= 256 / <= 272 points): lets the dense dispatch size its class launches exactly instead of P workgroups each
    const int32_t* hint_off = nullptr;
    int hint_P = 0, hint_le256 = 0, hint_le272 = 0;
    std::mutex mu;
    char err[512] = {0};
    const char* last_dense_kernel = "";
    // Ownership (include/gpc.h): every object created from the context (gpc_sparse, gpc_patches) holds one reference.
    // gpc_ctx_destroy releases the device resources and the caller's reference; the struct itself lives until the last
    // child has been destroyed, so a child destroyed AFTER its context finds a `dead` context instead of freed memory.
    std::atomic<int> refs{1};
    std::atomic<bool> dead{false};
};

static inline void gpc_ctx_ref(gpc_ctx* ctx) { ctx->refs.fetch_add(1, std::memory_order_relaxed); }
static inline void gpc_ctx_unref(gpc_ctx* ctx)
{
    if (ctx->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete ctx;
}

static inline int gpc_fail(gpc_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define GPC_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return gpc_fail((ctx), e_ == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "%s failed: %s",  \
                            #call, hipGetErrorString(e_));                                              \
    } while (0)

// The context's stream as the caller configured it, for code that does NOT hold ctx->mu: the dispatchers swap ctx->stream for the length of
// a fork (class streams, the host pipeline's compute streams) and put it back before they unlock, so a read under the lock never sees a
// fork's stream while a bare read from another thread can.
static inline hipStream_t gpc_stream_of(gpc_ctx* ctx)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ctx->stream;
}

// grow-only workspace; returns nullptr + sets error on failure.  Caller holds ctx->mu.
static inline int gpc_ws_reserve(gpc_ctx* ctx, size_t bytes)
{
    // "Thread-safe per context" (include/gpc.h) while a two-stream host-pointer call is in flight: its chunk kernels use the workspace from
    // the context's two compute streams, which the stream of ANOTHER thread's call (ws_len == 0: not a chunk of the pipeline) is not
    // ordered against.  Every user of the workspace comes through here first: that call goes behind the chunks already enqueued; the
    // chunks enqueued after it go behind the context's stream in turn (dense_dispatch).
    if (ctx->pipe_active && ctx->ws_len == 0 && !getenv("GPC_NO_PIPE_ORDER")) {   // (the switch: what the test of this looks like without it)
        ++ctx->foreign_gen;
        GPC_HIP(ctx, hipEventRecord(ctx->ev[0][10], ctx->own_stream));
        GPC_HIP(ctx, hipEventRecord(ctx->ev[1][10], ctx->s_c2));
        GPC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[0][10], 0));
        GPC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[1][10], 0));
    }
    if (bytes <= ctx->ws_bytes) return GPC_OK;
    // The new block is allocated BEFORE the old one is released: a request the device cannot serve leaves the context with the
    // workspace it had.  Only when old + new do not fit side by side is the old one given up first.
    void* nw = nullptr;
    // (diagnostic: GPC_WS_FAIL_ABOVE=<bytes> makes every larger request fail like an exhausted device -- the test of the fallbacks)
    const char* lim = getenv("GPC_WS_FAIL_ABOVE");
    if (lim && bytes > (size_t)atoll(lim)) return gpc_fail(ctx, GPC_ENOMEM, "workspace allocation failed (GPC_WS_FAIL_ABOVE)");
    hipError_t e = hipMalloc(&nw, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (!ctx->ws) return gpc_fail(ctx, GPC_ENOMEM, "workspace allocation failed");
        GPC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t old_bytes = ctx->ws_bytes;
        GPC_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        e = hipMalloc(&nw, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // put back what was there, so that later (smaller) calls find the context as it was
            if (hipMalloc(&ctx->ws, old_bytes) == hipSuccess) {
                ctx->ws_bytes = old_bytes;
                (void)hipMemsetAsync(ctx->ws, getenv("GPC_POISON_LDS") ? 0xFF : 0, old_bytes, ctx->stream);
            } else {
                (void)hipGetLastError();
                ctx->ws = nullptr;
            }
            return gpc_fail(ctx, GPC_ENOMEM, "workspace allocation failed");
        }
    } else if (ctx->ws) {
        // the previous workspace may still be in use by work enqueued on the stream
        GPC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GPC_HIP(ctx, hipFree(ctx->ws));
    }
    ctx->ws = nw;
    ctx->ws_bytes = bytes;
    // recycled device memory holds arbitrary bit patterns; the kernels write every workspace element before they use it,
    // but their (clamped, unconditional) prefetches may touch elements they never consume: keep those reads free of
    // signalling patterns and the results independent of what ran before
    GPC_HIP(ctx, hipMemsetAsync(ctx->ws, getenv("GPC_POISON_LDS") ? 0xFF : 0, bytes, ctx->stream));   // diagnostic runs: NaN instead of zero
    return GPC_OK;
}

// Diagnostic (GPC_POISON_LDS=1): before every kernel family runs, fill the LDS of every CU with NaN.  LDS keeps what the
// previous kernel left there, so a read of a word the current kernel never wrote is otherwise a coin toss; with the poison it
// is a NaN in the output and a failing test.  Enqueued on the context's stream; a no-op unless the variable is set.
extern "C" int gpc_debug_poison_lds(gpc_ctx* ctx);

// ---- launchers implemented in the kernel translation units -------------------------------------------------

struct DenseArgs {
    gpc_params prm;
    int P, n_max, n_total, ny, m;
    const int32_t* off;
    const double *x0, *x1, *y;
    const double *xs0, *xs1;   // point-wise X* (m entries) or nullptr when the grid form is used
    double grid_res;           // grid form: res, sz (m = sz*sz)
    int grid_sz;
    double *f_star, *v_star, *alpha_out;
    int32_t* status;
    // size-class dispatch of a ragged batch (gpc_api.hip): when sel != nullptr a kernel works on the patches sel[0 .. *sel_count)
    // (both on the device) instead of 0 .. P-1
    const int32_t* sel;
    const int32_t* sel_count;
    int sel_base;              // generic kernel only: it works on sel[sel_base .. *sel_count) -- the overflow launch behind a class launch that
                               // was sized from a host-side hint (gpc_api.hip)
};

// generic kernel: any n <= GPC_MAX_POINTS, K/L in a global-memory workspace slot per workgroup
size_t dense_generic_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out);
int dense_generic_launch(gpc_ctx* ctx, const DenseArgs& a, int grid, double* ws_override = nullptr);

// register-tile MFMA kernel: n <= 256, trailing matrix resident in VGPRs (see dense_mfma.hip)
bool dense_mfma_supported(const DenseArgs& a);
int dense_mfma_launch(gpc_ctx* ctx, const DenseArgs& a);

// predictive variance from the exported factor of the register-tile kernel (dense_variance.hip): V* [P][m]
int dense_variance_launch(gpc_ctx* ctx, const DenseArgs& a, int nt_max, const double* factor, const double* alpha, double* v_star);

// ... and from the tiled kernel's per-patch factor slots (n <= 1024); scratch: V blocks of the waves in flight
size_t big_slot_doubles(int ntw);
size_t dense_variance_big_scratch_doubles(const gpc_ctx* ctx, int ntw);
int dense_variance_big_launch(gpc_ctx* ctx, const DenseArgs& a, int ntw, const double* ws, size_t slot, const double* alpha,
                              double* scratch, double* v_star);

// tiled left-looking MFMA kernel: 256 < n <= 1024, factor in a global-memory workspace slot per workgroup (see dense_mfma_big.hip)
bool dense_big_supported(const DenseArgs& a);
size_t dense_big_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out);
int dense_big_launch(gpc_ctx* ctx, const DenseArgs& a, int grid);
// one wave per patch, eight patches per CU: n <= 256, depth plane, mean only (see dense_mfma_w1.hip) -- the C2 headline kernel
bool dense_w1_supported(const DenseArgs& a);
size_t dense_w1_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out, int cap = 0);
int dense_w1_launch(gpc_ctx* ctx, const DenseArgs& a, int grid);
// the same kernel inside the Newton / IRLS loop of the probit variant (BASELINE config 5; any n <= 1024, ny == 1)
struct IrlsArgs {
    int max_iter;
    double tol, f_init;
    int32_t* iters;   // [P] or nullptr
    double* fhat;     // [n_total] or nullptr
};
int dense_irls_launch(gpc_ctx* ctx, const DenseArgs& a, const IrlsArgs& ir, int grid);
