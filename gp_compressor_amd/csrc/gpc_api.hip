// gpc_api.hip -- C-ABI entry points of libgpc_hip.so: context, parameter defaults, dense-path dispatch,
// patch->rank partition.  See include/gpc.h for the contract and the reference call sites each one replaces.
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <condition_variable>
#include <thread>
#include <vector>

#include "gpc_device.h"
#include "gpc_internal.h"

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(gpc_ctx* ctx, size_t bytes)
    {
        if (bytes == 0) bytes = 8;
        GPC_HIP(ctx, hipMalloc(&p, bytes));
        return GPC_OK;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

int host_n_max(gpc_ctx* ctx, int P, const int32_t* off, int* n_max, int* n_total)
{
    if (P > 0 && off[0] != 0) return gpc_fail(ctx, GPC_EINVAL, "off[0] must be 0");
    int mx = 0;
    for (int i = 0; i < P; ++i) {
        int n = off[i + 1] - off[i];
        if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "off must be non-decreasing (patch %d)", i);
        mx = std::max(mx, n);
    }
    *n_max = mx;
    *n_total = P > 0 ? off[P] : 0;
    return GPC_OK;
}
}  // namespace

extern "C" {

int gpc_version(void) { return GPC_VERSION; }

// gaussian_process(double sigmaf = 0.05, double l = 3, double sigman = 0.04)  (/root/reference/src/gaussian_process.h:21),
// squared by the constructor (/root/reference/src/gaussian_process.cpp:8-9)
void gpc_default_params_dense(gpc_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->sigmaf_sq = 0.05 * 0.05;
    p->l_sq = 3.0 * 3.0;
    p->noise = 0.04 * 0.04;
    p->eps_tol = 0.0;
    p->capacity = 0;
    p->noise_model = 0;
    p->ref_double_noise = 1;
    p->ref_field_delete_bug = 1;
    p->want_variance = 0;
}

// sparse_gp(int capacity = 100, double s0 = 1e-1f), eps_tol(1e-6f)       (/root/reference/src/sparse_gp.h:48, sparse_gp.hpp:30)
// sparse_gp_field(int capacity = 100, double s0 = 1e2f), eps_tol(1e-4f)  (/root/reference/src/sparse_gp_field.h:43, .hpp:16)
// rbf_kernel(double sigmaf_sq = 100e-0f, double l_sq = 1*1)              (/root/reference/src/rbf_kernel.h:24)
void gpc_default_params_sparse(gpc_params* p, int ny)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->sigmaf_sq = (double)100e-0f;
    p->l_sq = 1 * 1;
    p->capacity = 100;
    p->noise_model = 0;
    p->ref_double_noise = 1;
    p->ref_field_delete_bug = 1;
    if (ny == 1) {
        p->noise = (double)1e-1f;
        p->eps_tol = (double)1e-6f;
    } else {
        p->noise = (double)1e2f;
        p->eps_tol = (double)1e-4f;
    }
}

static int gpc_aux_streams(gpc_ctx* ctx);
int gpc_ctx_create(gpc_ctx** out, int device)
{
    if (!out) return GPC_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GPC_ENODEV;   // no CPU fallback
    if (device < 0 || device >= count) return GPC_ENODEV;
    if (hipSetDevice(device) != hipSuccess) return GPC_ENODEV;
    gpc_ctx* ctx = new (std::nothrow) gpc_ctx();
    if (!ctx) return GPC_ENOMEM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return GPC_EHIP;
    }
    ctx->stream = ctx->own_stream;
    // The three auxiliary streams are created HERE, back to back with the context's own: the runtime deals streams onto its (four)
    // hardware queues in creation order, and two of ours on one queue serialise what the host-pointer pipeline wants to overlap -- created
    // lazily after another library (PyTorch) had made streams of its own, the second compute stream landed on the first one's queue and a
    // C2 call took 2.53 ms instead of 1.99 (GPU_MAX_HW_QUEUES=2 in a process of our own: 4.1 ms).
    if (gpc_aux_streams(ctx) != GPC_OK) {
        gpc_ctx_destroy(ctx);
        return GPC_EHIP;
    }
    *out = ctx;
    return GPC_OK;
}

int gpc_ctx_set_stream(gpc_ctx* ctx, void* hip_stream)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->stream = (hip_stream == GPC_STREAM_OWN) ? ctx->own_stream : static_cast<hipStream_t>(hip_stream);
    return GPC_OK;
}

int gpc_ctx_synchronize(gpc_ctx* ctx)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    GPC_HIP(ctx, hipStreamSynchronize(gpc_stream_of(ctx)));
    return GPC_OK;
}

int gpc_dev_malloc(gpc_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    GPC_HIP(ctx, hipMalloc(out, bytes ? bytes : 8));
    return GPC_OK;
}

int gpc_dev_free(gpc_ctx* ctx, void* p)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!p) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    GPC_HIP(ctx, hipStreamSynchronize(gpc_stream_of(ctx)));     // nothing enqueued may still be using it
    GPC_HIP(ctx, hipFree(p));
    return GPC_OK;
}

int gpc_host_alloc(gpc_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    GPC_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
    return GPC_OK;
}

int gpc_host_free(gpc_ctx* ctx, void* p)
{
    if (!ctx) return GPC_EINVAL;
    if (!p) return GPC_OK;
    GPC_HIP(ctx, hipHostFree(p));
    return GPC_OK;
}

int gpc_dev_memcpy(gpc_ctx* ctx, void* dst, const void* src, size_t bytes, int kind)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (bytes == 0) return GPC_OK;
    if (!dst || !src) return gpc_fail(ctx, GPC_EINVAL, "dst/src is NULL");
    const hipMemcpyKind k = kind == GPC_COPY_H2D ? hipMemcpyHostToDevice : kind == GPC_COPY_D2H ? hipMemcpyDeviceToHost
                          : kind == GPC_COPY_D2D ? hipMemcpyDeviceToDevice : hipMemcpyDefault;
    if (k == hipMemcpyDefault) return gpc_fail(ctx, GPC_EINVAL, "kind must be GPC_COPY_H2D, _D2H or _D2D");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = gpc_stream_of(ctx);
    GPC_HIP(ctx, hipMemcpyAsync(dst, src, bytes, k, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

// Children (gpc_sparse, gpc_patches) may outlive the context: the device resources go now, the struct when the last
// child is destroyed.  A child of a dead context can only be destroyed; every other call on it returns GPC_EINVAL.
void gpc_ctx_destroy(gpc_ctx* ctx)
{
    if (!ctx || ctx->dead.load()) return;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->ws) (void)hipFree(ctx->ws);
        ctx->ws = nullptr;
        if (ctx->tickets) (void)hipFree(ctx->tickets);
        ctx->tickets = nullptr;
        ctx->ws_bytes = 0;
        if (ctx->s_in) {
            (void)hipStreamSynchronize(ctx->s_in);
            (void)hipStreamSynchronize(ctx->s_out);
            (void)hipStreamDestroy(ctx->s_in);
            (void)hipStreamDestroy(ctx->s_out);
            if (ctx->s_c2) { (void)hipStreamSynchronize(ctx->s_c2); (void)hipStreamDestroy(ctx->s_c2); ctx->s_c2 = nullptr; }
            for (auto& row : ctx->ev)
                for (auto& e : row) if (e) (void)hipEventDestroy(e);
            ctx->s_in = ctx->s_out = nullptr;
        }
        if (ctx->io) (void)hipFree(ctx->io);
        if (ctx->pin_in) (void)hipHostFree(ctx->pin_in);
        if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
        ctx->io = ctx->pin_in = ctx->pin_out = nullptr;
        ctx->io_bytes = ctx->pin_in_bytes = ctx->pin_out_bytes = 0;
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        ctx->own_stream = nullptr;
        ctx->stream = nullptr;
        snprintf(ctx->err, sizeof(ctx->err), "the context has been destroyed");
        ctx->dead.store(true);
    }
    gpc_ctx_unref(ctx);
}

const char* gpc_last_error(const gpc_ctx* ctx) { return ctx ? ctx->err : "null context"; }
const char* gpc_last_dense_kernel(const gpc_ctx* ctx) { return ctx ? ctx->last_dense_kernel : ""; }

// ------------------------------------------------------------------------------------------------ diagnostics

__global__ __launch_bounds__(256) void gpc_poison_lds_kernel(int words)
{
    extern __shared__ unsigned long long lds_words[];
    for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = 0x7ff8dead0000beefull;      // a quiet NaN
    __syncthreads();
    // every workgroup takes a whole CU's LDS: hold it for a moment so that the first wave of workgroups covers all CUs
    for (int k = 0; k < 64; ++k) __builtin_amdgcn_s_sleep(127);
    if (lds_words[(threadIdx.x * 17) % words] == 0ull) lds_words[0] = 1ull;                  // keeps the stores alive
}

int gpc_debug_poison_lds(gpc_ctx* ctx)
{
    const bool on = getenv("GPC_POISON_LDS") != nullptr;       // read per call: a test can switch it on for itself
    if (!on || !ctx) return GPC_OK;
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gpc_poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    const int bytes = 160 * 1024;
    hipLaunchKernelGGL(gpc_poison_lds_kernel, dim3(ctx->num_cus * 2), dim3(256), bytes, ctx->stream, bytes / 8);
    GPC_HIP(ctx, hipGetLastError());
    // the device workspace is per-call scratch as well: whatever the previous call left in it becomes NaN (all-ones doubles)
    // (only the region of the launch in hand: the host-pointer pipeline runs another chunk's kernel in the other half at the same time)
    if (ctx->ws && ctx->ws_bytes > ctx->ws_off) {
        const size_t len = ctx->ws_len ? std::min(ctx->ws_len, ctx->ws_bytes - ctx->ws_off) : ctx->ws_bytes - ctx->ws_off;
        GPC_HIP(ctx, hipMemsetAsync(static_cast<char*>(ctx->ws) + ctx->ws_off, 0xFF, len, ctx->stream));
    }
    return GPC_OK;
}

// ------------------------------------------------------------------------------------------------ dense path

static int dense_check(gpc_ctx* ctx, const gpc_params* prm, int P, const void* off, int n_max, int n_total,
                       const void* x0, const void* x1, const void* y, int ny, int m, const void* f_star)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!prm) return gpc_fail(ctx, GPC_EINVAL, "params is NULL");
    if (P < 0 || m < 0 || n_total < 0 || n_max < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (ny != 1 && ny != 3) return gpc_fail(ctx, GPC_EINVAL, "ny must be 1 (depth) or 3 (RGB), got %d", ny);
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (n_total > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (P > 0 && m > 0 && !f_star) return gpc_fail(ctx, GPC_EINVAL, "f_star is NULL");
    if (n_max > GPC_MAX_POINTS) return gpc_fail(ctx, GPC_ERANGE, "n_max %d > GPC_MAX_POINTS %d", n_max, GPC_MAX_POINTS);
    if (!(prm->l_sq > 0.0) || !(prm->sigmaf_sq >= 0.0) || !(prm->noise >= 0.0))
        return gpc_fail(ctx, GPC_EINVAL, "kernel/noise parameters out of range");
    return GPC_OK;
}

// Ragged batches whose largest patch exceeds the register-resident kernel's 256 points: patches are sorted into size classes on
// the device (no host round trip: `off` lives there) -- n <= 256 -> the register-resident kernel, 256 < n <= 272 -> its NT = 17
// shape (depth plane only: the octree leaves of a cloud cut for 256-point patches scatter around that size, median 258, and
// would otherwise pay the tiled kernel's 4x cost per patch), the rest -> the tiled kernel.  Patches are independent, so the
// order inside a class does not matter.
__global__ void dense_classify_kernel(int P, const int32_t* off, int bound0, int bound1, int32_t* sel0, int32_t* sel1, int32_t* sel2,
                                      int32_t* counts)
{
    // one atomic per wave and class (a counter takes ~11 ns per atomic device-wide: with one per patch the 8192 patches of a
    // single-class batch spent 95 us here, 0.8 % of a C3 launch and 5 % of a C2-sized one)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int n = i < P ? off[i + 1] - off[i] : 0;
    const int cls = i >= P ? -1 : n <= bound0 ? 0 : n <= bound1 ? 1 : 2;
    int32_t* const sel[3] = {sel0, sel1, sel2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == c);
        if (m == 0) continue;
        int base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&counts[c], __builtin_popcountll(m));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
        if (cls == c) sel[c][base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
    }
}

// the context's two auxiliary streams and its events (also used by the host-pointer pipeline), created on first use
static int gpc_aux_streams(gpc_ctx* ctx)
{
    if (ctx->s_in) return GPC_OK;
    GPC_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
    GPC_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
    GPC_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_c2, hipStreamNonBlocking));
    for (auto& row : ctx->ev)
        for (auto& e : row) GPC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return GPC_OK;
}

// The host-pointer pipeline (dense_host) runs the kernels of consecutive chunks on TWO streams, each in its own half of the workspace, so
// that one chunk's last workgroups and the next chunk's first ones overlap: the stream and workspace offset of the call in hand, set by
// the calling thread around its _dev call and applied under the context lock.
static thread_local hipStream_t tl_stream_override = nullptr;
static thread_local size_t tl_ws_off = 0, tl_ws_len = 0;
static thread_local unsigned tl_seen_gen[2] = {0, 0};   // gpc_ctx::foreign_gen as the pipeline's two compute streams last saw it

// the one-wave kernel takes this batch (the rule of dense_dispatch_locked, also asked by dense_host before it splits a batch over two streams)
static bool dense_w1_takes(const gpc_ctx* ctx, const DenseArgs& a)
{
    const char* mp = getenv("GPC_W1_MIN_P");
    const int min_p = mp ? atoi(mp) : 4 * ctx->num_cus;
    const bool force = getenv("GPC_FORCE_GENERIC") || getenv("GPC_FORCE_BIG");
    // (round 4: the kernel's 512-point instance takes the depth plane of batches whose largest patch has 257 .. 512 points -- C3, and the
    // ragged batches of a cloud cut for 256-point patches, which the size-class split used to deal to three kernels; GPC_NO_W1_512=1: as before)
    if (a.n_max > 256 && getenv("GPC_NO_W1_512")) return false;
    return !force && a.n_max <= 512 && a.P >= min_p && a.P > 1 && (a.n_max > 192 || !a.v_star) && dense_w1_supported(a) && !getenv("GPC_NO_W1");
}

static int dense_dispatch_locked(gpc_ctx* ctx, DenseArgs& a);
static int dense_dispatch(gpc_ctx* ctx, DenseArgs& a)
{
    if (a.P == 0 || (a.m == 0 && !a.alpha_out)) return GPC_OK;
    if (a.n_max < 1) a.n_max = 1;
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t const saved = ctx->stream;
    if (tl_stream_override) {
        // a chunk of the two-stream host pipeline.  If a call of another thread touched the workspace since this compute stream's last
        // chunk (gpc_ws_reserve counts them and orders them behind the chunks enqueued before), the chunk goes behind the context's stream
        // as it stands now: that call may use -- or re-grow and clear -- the region this chunk is about to write.
        unsigned& seen = tl_seen_gen[tl_ws_off ? 1 : 0];
        if (seen != ctx->foreign_gen) {
            seen = ctx->foreign_gen;
            if (saved != tl_stream_override) {
                GPC_HIP(ctx, hipEventRecord(ctx->ev[2][10], saved));
                GPC_HIP(ctx, hipStreamWaitEvent(tl_stream_override, ctx->ev[2][10], 0));
            }
        }
        ctx->stream = tl_stream_override;
    }
    ctx->ws_off = tl_ws_off;
    ctx->ws_len = tl_ws_len;
    const int rc = dense_dispatch_locked(ctx, a);
    ctx->stream = saved;
    ctx->ws_off = 0;
    ctx->ws_len = 0;
    return rc;
}

static int dense_dispatch_locked(gpc_ctx* ctx, DenseArgs& a)
{
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (!a.prm.want_variance) a.v_star = nullptr;
    {
        const int rcp = gpc_debug_poison_lds(ctx);
        if (rcp != GPC_OK) return rcp;
    }
    const bool force_generic = getenv("GPC_FORCE_GENERIC") != nullptr, force_big = getenv("GPC_FORCE_BIG") != nullptr;
    const bool ny_ok = a.ny == 1 || a.ny == 3;
    if (!force_generic && !force_big && ny_ok) {
        const bool no_split = a.P == 1 || getenv("GPC_NO_SPLIT");
        // Depth plane, n <= 256, a batch large enough to fill the chip: ONE wave per patch, eight patches per CU (dense_mfma_w1.hip) --
        // no hand-over between waves at all.  Measured against the register-resident kernel at 8192 patches: 256 points 1.72 against
        // 2.57 ms, 192: 0.98 / 1.70, 128: 0.48 / 1.06, 64: 0.20 / 0.62; at 512 patches the two are level, below that the register kernel's
        // eight waves per patch win on latency (64 patches x 192 points: 0.066 against 0.134 ms) -- hence the batch-size rule
        // (GPC_W1_MIN_P overrides it; the variance goes this way for 193 .. 256 points, where its solve kernel is the <16> shape).
        {
            if (dense_w1_takes(ctx, a)) {
                // One factor slot (304 KB) per patch of a launch.  If the device cannot serve that, the batch goes
                // through in smaller launches that reuse fewer slots, and below two launches' worth of resident patches it takes the
                // register-resident kernel, which needs no workspace at all.
                int grid_w1 = 0, rcw = GPC_ENOMEM;
                DenseArgs aw = a;
                for (int cap_w1 = 0;;) {
                    const size_t w1_bytes = (dense_w1_ws_bytes(ctx, aw, &grid_w1, cap_w1) + 255) & ~(size_t)255;
                    rcw = gpc_ws_reserve(ctx, ctx->ws_off + w1_bytes);
                    if (rcw != GPC_ENOMEM) break;
                    cap_w1 = grid_w1 / 2;
                    if (cap_w1 < 4 * ctx->num_cus) break;             // (below the batch-size rule of this kernel)
                }
                if (rcw == GPC_OK) return dense_w1_launch(ctx, a, grid_w1);
                if (rcw != GPC_ENOMEM || a.v_star) return rcw;
                (void)hipGetLastError();
            }
        }
        // (GPC_W2=1: the two-wave shape of the tiled kernel, round 3's first headline kernel, kept as a cross-check)
        if (a.n_max > 192 && a.n_max <= 256 && a.ny == 1 && !a.v_star && a.P > 1 && getenv("GPC_W2")) {
            int grid_w2 = 0;
            const size_t w2_bytes = (dense_big_ws_bytes(ctx, a, &grid_w2) + 255) & ~(size_t)255;
            const int rcw = gpc_ws_reserve(ctx, w2_bytes);
            if (rcw != GPC_OK) return rcw;
            return dense_big_launch(ctx, a, grid_w2);
        }
        if (a.n_max <= 256 || (no_split && dense_mfma_supported(a) && !getenv("GPC_NO_NT17")))
            return dense_mfma_launch(ctx, a);                                                               // one shape for the whole batch
        // (a batch whose patches all have n_max points -- P n_max == n_total: every n_i <= n_max and they add up to n_total -- has one
        // size class and the host knows it: no classification, no empty class launches waiting for a CU beside the tiled kernel)
        const bool uniform = (long long)a.P * a.n_max == (long long)a.n_total && !getenv("GPC_NO_UNIFORM");
        if (!a.v_star && a.n_max <= GPC_MAX_POINTS && !no_split && !(uniform && a.n_max > 17 * 16)) {
            const bool nt17 = a.ny == 1 && !getenv("GPC_NO_NT17");
            const bool need_big = !(nt17 && a.n_max <= 17 * 16);
            int grid_b = 0;
            const size_t big_bytes = need_big ? (dense_big_ws_bytes(ctx, a, &grid_b) + 255) & ~(size_t)255 : 0;
            // Class sizes known on the host (the batch came from gpc_project_cloud on this context): the class launches get exactly that
            // many workgroups.  The hint is matched by pointer and P only -- a caller that rewrites `off` IN PLACE keeps both -- so every
            // hinted launch is followed by an OVERFLOW launch of the generic kernel (a few workgroups that stride over
            // sel[hint .. count), count on the device: no patch when the hint was right, every patch the hint missed otherwise).
            const bool hinted = ctx->hint_off == a.off && ctx->hint_P == a.P && !getenv("GPC_NO_HINT");
            constexpr int OVF_GRID = 8;
            const size_t ovf_slot = sizeof(double) * (size_t)(17 * 16 + a.ny) * (size_t)(17 * 16 + a.ny);
            const size_t ovf_bytes = hinted ? (2 * OVF_GRID * ovf_slot + 255) & ~(size_t)255 : 0;
            int rc = gpc_ws_reserve(ctx, big_bytes + ovf_bytes + sizeof(int32_t) * (3 * (size_t)a.P + 64));
            if (rc != GPC_OK) return rc;
            double* ovf_ws = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + big_bytes);
            int32_t* counts = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->ws) + big_bytes + ovf_bytes);
            int32_t* sel0 = counts + 64;
            int32_t* sel1 = sel0 + a.P;
            int32_t* sel2 = sel1 + a.P;
            GPC_HIP(ctx, hipMemsetAsync(counts, 0, 3 * sizeof(int32_t), ctx->stream));
            hipLaunchKernelGGL(dense_classify_kernel, dim3((a.P + 255) / 256), dim3(256), 0, ctx->stream, a.P, a.off, 256, nt17 ? 17 * 16 : 256,
                               sel0, sel1, sel2, counts);
            GPC_HIP(ctx, hipGetLastError());
            // The classes are independent: they run on three streams (forked from and joined to the context's), so that
            // the tail of one kernel -- its last workgroups running on a mostly idle chip -- fills with the next one's work.
            const bool fork = !getenv("GPC_NO_FORK");
            hipStream_t main_s = ctx->stream;
            bool forked = false;
            // an error after the fork must not leave work (or the caller's buffers) in flight on the side streams
            auto bail = [&](int code) {
                ctx->stream = main_s;
                if (forked) { (void)hipStreamSynchronize(ctx->s_in); (void)hipStreamSynchronize(ctx->s_out); }
                return code;
            };
            if (fork) {
                rc = gpc_aux_streams(ctx);
                if (rc != GPC_OK) return rc;
                GPC_HIP(ctx, hipEventRecord(ctx->ev[0][13], main_s));
                GPC_HIP(ctx, hipStreamWaitEvent(ctx->s_in, ctx->ev[0][13], 0));
                GPC_HIP(ctx, hipStreamWaitEvent(ctx->s_out, ctx->ev[0][13], 0));
                forked = true;
            }
            // largest patches first: the tiled kernel's workgroups are the long ones
            if (need_big) {
                DenseArgs b = a;
                b.sel = sel2; b.sel_count = counts + 2;
                if (fork) ctx->stream = ctx->s_out;
                rc = dense_big_launch(ctx, b, grid_b);
                ctx->stream = main_s;
                if (rc != GPC_OK) return bail(rc);
            }
            // Without a hint: P workgroups per class -- the ones beyond the class count leave at once, but each still waits for a CU
            // with 158 KB of LDS free.
            auto overflow = [&](const DenseArgs& cls, int launched, int which) -> int {
                if (!hinted || launched >= a.P) return GPC_OK;
                DenseArgs o = cls;
                o.sel_base = launched;
                o.P = a.P;
                return dense_generic_launch(ctx, o, OVF_GRID, ovf_ws + (size_t)which * OVF_GRID * (ovf_slot / sizeof(double)));
            };
            const int c0 = hinted ? ctx->hint_le256 : a.P;
            const int c1 = hinted ? (nt17 ? ctx->hint_le272 - ctx->hint_le256 : 0) : a.P;
            DenseArgs s = a;
            if (nt17 && c1 > 0) {
                s.n_max = 17 * 16;
                s.sel = sel1; s.sel_count = counts + 1;
                s.P = c1;
                if (fork) ctx->stream = ctx->s_in;
                rc = dense_mfma_launch(ctx, s);
                if (rc == GPC_OK) rc = overflow(s, c1, 1);
                ctx->stream = main_s;
                if (rc != GPC_OK) return bail(rc);
            } else if (nt17 && hinted) {
                s.n_max = 17 * 16;
                s.sel = sel1; s.sel_count = counts + 1;
                rc = overflow(s, 0, 1);
                if (rc != GPC_OK) return bail(rc);
            }
            if (c0 > 0) {
                s.n_max = 256;
                s.sel = sel0; s.sel_count = counts;
                s.P = c0;
                rc = dense_mfma_launch(ctx, s);
                if (rc == GPC_OK) rc = overflow(s, c0, 0);
                if (rc != GPC_OK) return bail(rc);
            } else if (hinted) {
                s.n_max = 256;
                s.sel = sel0; s.sel_count = counts;
                rc = overflow(s, 0, 0);
                if (rc != GPC_OK) return bail(rc);
            }
            if (fork) {
                GPC_HIP(ctx, hipEventRecord(ctx->ev[1][13], ctx->s_in));
                GPC_HIP(ctx, hipEventRecord(ctx->ev[2][13], ctx->s_out));
                GPC_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev[1][13], 0));
                GPC_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev[2][13], 0));
            }
            ctx->last_dense_kernel = !need_big ? "dense_mfma_nt16 + dense_mfma_nt17" : nt17 ? "dense_mfma_nt16 + dense_mfma_nt17 + dense_mfma_big"
                                                                                             : "dense_mfma_nt16 + dense_mfma_big";
            return rc;
        }
    }
    if ((dense_big_supported(a) || (force_big && a.n_max <= 1024 && !a.v_star)) && !force_generic) {   // GPC_FORCE_BIG: diagnostic
        int grid_b = 0;
        const size_t big_bytes = (dense_big_ws_bytes(ctx, a, &grid_b) + 255) & ~(size_t)255;
        int rcb = gpc_ws_reserve(ctx, big_bytes);
        if (rcb != GPC_OK) return rcb;
        return dense_big_launch(ctx, a, grid_b);
    }
    int grid = 0;
    size_t bytes = dense_generic_ws_bytes(ctx, a, &grid);
    int rc = gpc_ws_reserve(ctx, bytes);
    if (rc != GPC_OK) return rc;
    return dense_generic_launch(ctx, a, grid);
}

int gpc_dense_fit_predict_dev(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off, int n_max, int n_total,
                              const double* x0, const double* x1, const double* y, int ny,
                              int m, const double* xs0, const double* xs1,
                              double* f_star, double* v_star, double* alpha_out, int32_t* status)
{
    int rc = dense_check(ctx, params, P, off, n_max, n_total, x0, x1, y, ny, m, f_star);
    if (rc != GPC_OK) return rc;
    if (m > 0 && (!xs0 || !xs1)) return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1 is NULL");
    DenseArgs a{};
    a.prm = *params;
    a.P = P; a.n_max = n_max; a.n_total = n_total; a.ny = ny; a.m = m;
    a.off = off; a.x0 = x0; a.x1 = x1; a.y = y; a.xs0 = xs0; a.xs1 = xs1;
    a.grid_res = 0.0; a.grid_sz = 0;
    a.f_star = f_star; a.v_star = v_star; a.alpha_out = alpha_out; a.status = status;
    return dense_dispatch(ctx, a);
}

int gpc_dense_fit_predict_grid_dev(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off, int n_max,
                                   int n_total, const double* x0, const double* x1, const double* y, int ny,
                                   double res, int sz, double* f_star, double* alpha_out, int32_t* status)
{
    if (ctx && (sz < 0 || sz > 1024)) return gpc_fail(ctx, GPC_EINVAL, "sz out of range");
    int rc = dense_check(ctx, params, P, off, n_max, n_total, x0, x1, y, ny, sz * sz, f_star);
    if (rc != GPC_OK) return rc;
    DenseArgs a{};
    a.prm = *params;
    a.prm.want_variance = 0;
    a.P = P; a.n_max = n_max; a.n_total = n_total; a.ny = ny; a.m = sz * sz;
    a.off = off; a.x0 = x0; a.x1 = x1; a.y = y; a.xs0 = nullptr; a.xs1 = nullptr;
    a.grid_res = res; a.grid_sz = sz;
    a.f_star = f_star; a.v_star = nullptr; a.alpha_out = alpha_out; a.status = status;
    return dense_dispatch(ctx, a);
}

// ---- host-pointer wrappers: H2D, launch, D2H, synchronous -------------------------------------------------


// Copy with several threads: staging pageable caller memory through pinned buffers is a CPU memcpy, and one core moves
// ~10 GB/s where PCIe 5 moves 50.  A small pool owned by the process (created at the first host-pointer call on pageable
// memory, joined at exit) splits every copy into one slice per thread.
namespace {
class CopyPool {
public:
    static CopyPool& get()
    {
        static CopyPool p;
        return p;
    }
    void copy(void* dst, const void* src, size_t bytes)
    {
        if (bytes < (1u << 20) || th_.empty()) { std::memcpy(dst, src, bytes); return; }
        // One request at a time: the pool is process-wide while the callers' lock (ctx->host_mu) is per context, so two threads on
        // two contexts do get here together; the request fields below are shared with the workers.
        std::lock_guard<std::mutex> call(call_mu_);
        std::unique_lock<std::mutex> lk(m_);
        dst_ = (char*)dst; src_ = (const char*)src; bytes_ = bytes;
        remaining_ = (int)th_.size();
        ++gen_;
        lk.unlock();
        cv_.notify_all();
        slice(0);                                   // the caller takes slice 0
        lk.lock();
        done_.wait(lk, [&] { return remaining_ == 0; });
    }
private:
    CopyPool()
    {
        unsigned hc = std::thread::hardware_concurrency();
        int n = (int)std::min(4u, hc > 2 ? hc / 2 : 1u);      // measured on the 16-CPU share of a 1-GPU box: 4 threads 4.35 ms per C2 call, 8: 6.6, 12: 4.4
        if (const char* e = getenv("GPC_COPY_THREADS")) n = std::max(1, atoi(e));
        parts_ = n;
        for (int t = 1; t < n; ++t) th_.emplace_back([this, t] { run(t); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void slice(int t)
    {
        const size_t per = ((bytes_ / parts_) + 63) & ~(size_t)63;
        const size_t lo = std::min(bytes_, per * t), hi = (t == parts_ - 1) ? bytes_ : std::min(bytes_, per * (t + 1));
        if (lo < hi) std::memcpy(dst_ + lo, src_ + lo, hi - lo);
    }
    void run(int t)
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            lk.unlock();
            slice(t);
            lk.lock();
            if (--remaining_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_, call_mu_;
    std::condition_variable cv_, done_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0;
    int parts_ = 1, remaining_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};
}  // namespace
static void par_memcpy(void* dst, const void* src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }
extern "C" void gpc_test_par_memcpy(void* dst, const void* src, size_t bytes) { par_memcpy(dst, src, bytes); }   // host-only test hook

static bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int grow(gpc_ctx* ctx, void** p, size_t* have, size_t need, bool pinned)
{
    if (need <= *have) return GPC_OK;
    if (*p) {
        GPC_HIP(ctx, hipDeviceSynchronize());
        if (pinned) GPC_HIP(ctx, hipHostFree(*p)); else GPC_HIP(ctx, hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    need = (need + (need >> 2) + 4095) & ~(size_t)4095;           // 25 % head-room: ragged batches of one cloud vary a little
    if (pinned) GPC_HIP(ctx, hipHostMalloc(p, need, hipHostMallocDefault)); else GPC_HIP(ctx, hipMalloc(p, need));
    *have = need;
    return GPC_OK;
}

// Host-pointer entry of the dense path: H2D, kernel, D2H, synchronous for the caller -- but pipelined inside.  The batch is cut
// into up to four chunks of whole patches; chunk c+1 goes up (copy stream, SDMA engine) and chunk c-1 comes down (second copy
// stream) while the kernel runs on chunk c.  Pinned caller memory (gpc_host_alloc) is transferred in place; pageable memory is
// staged through the context's pinned buffers by a threaded memcpy, which overlaps the GPU work of the previous chunk as well.
// (Copies issued from pageable memory run as blit kernels that queue behind a compute kernel filling every CU: with those,
// chunking overlaps nothing -- measured in round 1.)
static int dense_host(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off,
                      const double* x0, const double* x1, const double* y, int ny,
                      int m, const double* xs0, const double* xs1, double res, int sz, bool grid,
                      double* f_star, double* v_star, double* alpha_out, int32_t* status)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (P < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    int n_max = 0, n_total = 0;
    int rc = host_n_max(ctx, P, off, &n_max, &n_total);
    if (rc != GPC_OK) return rc;
    rc = dense_check(ctx, params, P, off, n_max, n_total, x0, x1, y, ny, m, f_star);
    if (rc != GPC_OK) return rc;
    if (!grid && m > 0 && (!xs0 || !xs1)) return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1 is NULL");
    if (P == 0) return GPC_OK;
    std::lock_guard<std::mutex> hlk(ctx->host_mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const bool want_v = !grid && params->want_variance && v_star;
    const size_t N = (size_t)n_total;
    // (chunks of at least 1024 patches: below four patches per CU the dense dispatch leaves the one-wave-per-patch kernel)
    int C = getenv("GPC_HOST_NO_PIPELINE") ? 1 : P >= 4096 ? 4 : P >= 2048 ? 2 : 1;
    // Round 4: when the one-wave kernel takes the chunks, the kernels of consecutive chunks run on TWO streams, each in its own half of
    // the workspace (one factor slot per patch of a chunk), so that a chunk's draining workgroups and the next chunk's first ones share
    // the chip -- a chunk of 1024 .. 2048 patches is a single round of resident workgroups, i.e. all ramp and tail -- and the batch goes
    // through in EIGHT chunks: the first upload and the last download, which nothing overlaps, halve.  GPC_HOST_ONE_STREAM=1: as before.
    bool two = false;
    size_t half = 0;
    struct PipeFlag {           // gpc_ctx::pipe_active for the duration of a two-stream call, whichever way it ends
        gpc_ctx* c;
        bool on;
        ~PipeFlag()
        {
            if (!on) return;
            std::lock_guard<std::mutex> lk(c->mu);
            c->pipe_active = false;
        }
    } pipe{ctx, false};
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if ((rc = gpc_aux_streams(ctx))) return rc;
        DenseArgs probe{};
        probe.prm = *params;
        probe.P = P / 8; probe.n_max = n_max; probe.ny = ny; probe.m = m;
        probe.n_total = n_total;                       // (upper bound of a chunk's: the variance path keeps one weight per point in its half)
        probe.v_star = want_v ? v_star : nullptr;
        probe.xs0 = grid ? nullptr : xs0;
        // (measured on the C2 batch, same box: 1.95 against 2.12 ms per call, 4.2 against 3.87 M patches/s PCIe-inclusive; at 128 points per
        // patch the kernel is a third of the call and eight chunks only add transfers' fixed costs -- 1.09 against 0.96 ms -- hence n_max > 160)
        // EVERY chunk must go to the one-wave kernel: the other kernels know nothing of workspace halves (a chunk of small patches with
        // the variance wanted goes to the register kernel, whose factor export starts at the base of the workspace)
        bool all_w1 = C == 4 && P >= 8192 && n_max > 160 && !alpha_out && !getenv("GPC_HOST_ONE_STREAM") && ctx->own_stream != nullptr;
        for (int c = 0; c < 8 && all_w1; ++c) {
            const int p0 = (int)((long long)P * c / 8), p1 = (int)((long long)P * (c + 1) / 8);
            int nm = 1;
            for (int i = p0; i < p1; ++i) nm = std::max(nm, off[i + 1] - off[i]);
            DenseArgs pc = probe;
            pc.P = p1 - p0;
            pc.n_max = nm;
            all_w1 = dense_w1_takes(ctx, pc);
        }
        if (all_w1) {
            probe.P = (P + 7) / 8;
            half = (dense_w1_ws_bytes(ctx, probe, nullptr) + 255) & ~(size_t)255;
            if (gpc_ws_reserve(ctx, 2 * half) == GPC_OK) {
                two = true;
                C = 8;
                pipe.on = ctx->pipe_active = true;                 // (calls of other threads on this context: see gpc_ws_reserve)
                tl_seen_gen[0] = tl_seen_gen[1] = ctx->foreign_gen;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    // device arena: [off chunks | x0 | x1 | y planes per chunk | xs0 xs1 | f | v | alpha | status]
    auto al256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_off = al256(sizeof(int32_t) * (size_t)(P + C)), b_x = al256(8 * N), b_y = al256(8 * N * ny), b_xs = al256(8 * (size_t)m),
                 b_f = al256(8 * (size_t)P * ny * m), b_v = want_v ? al256(8 * (size_t)P * m) : 0, b_al = alpha_out ? al256(8 * N * ny) : 0,
                 b_st = al256(sizeof(int32_t) * (size_t)P);
    if ((rc = grow(ctx, &ctx->io, &ctx->io_bytes, b_off + 2 * b_x + b_y + 2 * b_xs + b_f + b_v + b_al + b_st, false))) return rc;
    char* d = static_cast<char*>(ctx->io);
    int32_t* d_off = (int32_t*)d; d += b_off;
    double* d_x0 = (double*)d; d += b_x;
    double* d_x1 = (double*)d; d += b_x;
    double* d_y = (double*)d; d += b_y;
    double* d_xs0 = (double*)d; d += b_xs;
    double* d_xs1 = (double*)d; d += b_xs;
    double* d_f = (double*)d; d += b_f;
    double* d_v = (double*)d; d += b_v;
    double* d_al = (double*)d; d += b_al;
    int32_t* d_st = (int32_t*)d;
    // pinned staging: inputs [off chunks | x0 | x1 | y] and outputs [f | v | alpha | status] -- only what is pageable on the caller's side
    const bool pin_x = is_pinned(x0) && is_pinned(x1) && is_pinned(y), pin_f = (m == 0 || is_pinned(f_star)) && (!want_v || is_pinned(v_star));
    if ((rc = grow(ctx, &ctx->pin_in, &ctx->pin_in_bytes, b_off + (pin_x ? 0 : 2 * b_x + b_y), true))) return rc;
    if ((rc = grow(ctx, &ctx->pin_out, &ctx->pin_out_bytes, b_st + (pin_f ? 0 : b_f + b_v), true))) return rc;
    char* hp = static_cast<char*>(ctx->pin_in);
    int32_t* h_off = (int32_t*)hp; hp += b_off;
    double* h_x0 = (double*)hp; hp += pin_x ? 0 : b_x;
    double* h_x1 = (double*)hp; hp += pin_x ? 0 : b_x;
    double* h_y = (double*)hp;
    char* ho = static_cast<char*>(ctx->pin_out);
    int32_t* h_st = (int32_t*)ho; ho += b_st;
    double* h_f = (double*)ho; ho += pin_f ? 0 : b_f;
    double* h_v = (double*)ho;
    hipStream_t sc, si = ctx->s_in, so = ctx->s_out;
    {   // dense_dispatch swaps ctx->stream for a moment while it forks (under ctx->mu): never read it half-way
        std::lock_guard<std::mutex> lk(ctx->mu);
        sc = ctx->stream;
    }
    // the arena may still be read by work a previous call left on the compute stream
    GPC_HIP(ctx, hipEventRecord(ctx->ev[0][15], sc));
    GPC_HIP(ctx, hipStreamWaitEvent(si, ctx->ev[0][15], 0));
    if (!grid && m) {
        GPC_HIP(ctx, hipMemcpyAsync(d_xs0, xs0, 8 * (size_t)m, hipMemcpyHostToDevice, si));
        GPC_HIP(ctx, hipMemcpyAsync(d_xs1, xs1, 8 * (size_t)m, hipMemcpyHostToDevice, si));
    }
    int p_lo[9];
    for (int c = 0; c <= C; ++c) p_lo[c] = (int)((long long)P * c / C);
    const hipStream_t sc_main = sc;
    // Both compute streams of the two-stream mode are the context's own (the call is synchronous for the caller anyway; its stream is
    // ordered in front of and behind them with events).  Two reasons, both measured: the legacy default stream does not overlap its
    // kernels with another stream's (2.49 against 1.98 ms per C2 call), and HIP deals streams onto its four hardware queues in creation
    // order, so a caller's stream made before the context can share a queue with s_c2 and serialise the pair (2.46 against 2.0 ms with
    // the bench on a torch side stream) -- own_stream, s_in, s_out and s_c2 are created back to back and never share one.
    const hipStream_t sc_a = two ? ctx->own_stream : sc_main;
    int fail = GPC_OK;
    for (int c = 0; c < C && fail == GPC_OK; ++c) {
        const int p0 = p_lo[c], Pc = p_lo[c + 1] - p0;
        const size_t r0 = (size_t)off[p0], Nc = (size_t)off[p0 + Pc] - r0;
        int32_t* ho_c = h_off + p0 + c;                                   // chunk c owns Pc + 1 entries
        int nmax_c = 0;
        for (int i = 0; i <= Pc; ++i) ho_c[i] = off[p0 + i] - off[p0];
        for (int i = 0; i < Pc; ++i) nmax_c = std::max(nmax_c, ho_c[i + 1] - ho_c[i]);
        int32_t* d_off_c = d_off + p0 + c;
        GPC_HIP(ctx, hipMemcpyAsync(d_off_c, ho_c, sizeof(int32_t) * (size_t)(Pc + 1), hipMemcpyHostToDevice, si));
        // chunk-local layout on the device: x0 | x1 | ny planes of Nc (the kernel's plane stride is the chunk's n_total)
        double* dx0 = d_x0 + r0; double* dx1 = d_x1 + r0; double* dy = d_y + r0 * ny;
        if (Nc) {
            const double *sx0 = x0 + r0, *sx1 = x1 + r0;
            if (!pin_x) {
                par_memcpy(h_x0 + r0, sx0, 8 * Nc);
                par_memcpy(h_x1 + r0, sx1, 8 * Nc);
                sx0 = h_x0 + r0; sx1 = h_x1 + r0;
            }
            GPC_HIP(ctx, hipMemcpyAsync(dx0, sx0, 8 * Nc, hipMemcpyHostToDevice, si));
            GPC_HIP(ctx, hipMemcpyAsync(dx1, sx1, 8 * Nc, hipMemcpyHostToDevice, si));
            for (int q = 0; q < ny; ++q) {
                const double* sy = y + (size_t)q * N + r0;
                if (!pin_x) { par_memcpy(h_y + r0 * ny + q * Nc, sy, 8 * Nc); sy = h_y + r0 * ny + q * Nc; }
                GPC_HIP(ctx, hipMemcpyAsync(dy + q * Nc, sy, 8 * Nc, hipMemcpyHostToDevice, si));
            }
        }
        GPC_HIP(ctx, hipEventRecord(ctx->ev[0][c], si));
        sc = (two && (c & 1)) ? ctx->s_c2 : sc_a;                        // this chunk's compute stream
        if (two && (c == 1 || (c == 0 && sc_a != sc_main))) {              // a stream of ours starts behind whatever the caller's stream carried
            GPC_HIP(ctx, hipStreamWaitEvent(sc, ctx->ev[0][15], 0));
            if (!grid && m && c == 1) GPC_HIP(ctx, hipStreamWaitEvent(sc, ctx->ev[0][0], 0));   // (xs0 / xs1 went up in front of chunk 0)
        }
        GPC_HIP(ctx, hipStreamWaitEvent(sc, ctx->ev[0][c], 0));
        double* df = d_f + (size_t)p0 * ny * m;
        tl_stream_override = two ? sc : nullptr;
        tl_ws_off = (two && (c & 1)) ? half : 0;
        tl_ws_len = two ? half : 0;
        if (grid)
            rc = gpc_dense_fit_predict_grid_dev(ctx, params, Pc, d_off_c, nmax_c, (int)Nc, dx0, dx1, dy, ny, res, sz, df,
                                                alpha_out ? d_al + r0 * ny : nullptr, d_st + p0);
        else
            rc = gpc_dense_fit_predict_dev(ctx, params, Pc, d_off_c, nmax_c, (int)Nc, dx0, dx1, dy, ny, m, d_xs0, d_xs1, df,
                                           want_v ? d_v + (size_t)p0 * m : nullptr, alpha_out ? d_al + r0 * ny : nullptr, d_st + p0);
        tl_stream_override = nullptr;
        tl_ws_off = 0;
        tl_ws_len = 0;
        if (rc != GPC_OK) { fail = rc; break; }
        GPC_HIP(ctx, hipEventRecord(ctx->ev[1][c], sc));
        GPC_HIP(ctx, hipStreamWaitEvent(so, ctx->ev[1][c], 0));
        if (m) GPC_HIP(ctx, hipMemcpyAsync(pin_f ? (void*)(f_star + (size_t)p0 * ny * m) : (void*)(h_f + (size_t)p0 * ny * m), df,
                                           8 * (size_t)Pc * ny * m, hipMemcpyDeviceToHost, so));
        if (want_v && m) GPC_HIP(ctx, hipMemcpyAsync(pin_f ? (void*)(v_star + (size_t)p0 * m) : (void*)(h_v + (size_t)p0 * m), d_v + (size_t)p0 * m,
                                                     8 * (size_t)Pc * m, hipMemcpyDeviceToHost, so));
        GPC_HIP(ctx, hipMemcpyAsync(h_st + p0, d_st + p0, sizeof(int32_t) * (size_t)Pc, hipMemcpyDeviceToHost, so));
        GPC_HIP(ctx, hipEventRecord(ctx->ev[2][c], so));
    }
    if (fail != GPC_OK) {
        (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(sc_main); (void)hipStreamSynchronize(so);
        if (two) { (void)hipStreamSynchronize(ctx->s_c2); (void)hipStreamSynchronize(sc_a); }
        return fail;
    }
    // alpha has chunk-local planes on the device ([ny][Nc] per chunk): gathered plane by plane at the end (rarely requested)
    for (int c = 0; c < C; ++c) {
        const int p0 = p_lo[c], Pc = p_lo[c + 1] - p0;
        GPC_HIP(ctx, hipEventSynchronize(ctx->ev[2][c]));
        if (!pin_f && m) par_memcpy(f_star + (size_t)p0 * ny * m, h_f + (size_t)p0 * ny * m, 8 * (size_t)Pc * ny * m);
        if (!pin_f && want_v && m) par_memcpy(v_star + (size_t)p0 * m, h_v + (size_t)p0 * m, 8 * (size_t)Pc * m);
        if (status) std::memcpy(status + p0, h_st + p0, sizeof(int32_t) * (size_t)Pc);
    }
    if (alpha_out && N) {
        for (int c = 0; c < C; ++c) {
            const int p0 = p_lo[c], Pc = p_lo[c + 1] - p0;
            const size_t r0 = (size_t)off[p0], Nc = (size_t)off[p0 + Pc] - r0;
            for (int q = 0; q < ny && Nc; ++q)
                GPC_HIP(ctx, hipMemcpyAsync(alpha_out + (size_t)q * N + r0, d_al + r0 * ny + q * Nc, 8 * Nc, hipMemcpyDeviceToHost, so));
        }
        GPC_HIP(ctx, hipStreamSynchronize(so));
    }
    if (two) {
        // the caller's stream is ordered behind our compute streams (the next _dev call on the context may reuse the workspace)
        GPC_HIP(ctx, hipEventRecord(ctx->ev[1][15], ctx->s_c2));
        GPC_HIP(ctx, hipStreamWaitEvent(sc_main, ctx->ev[1][15], 0));
        if (sc_a != sc_main) {
            GPC_HIP(ctx, hipEventRecord(ctx->ev[2][15], sc_a));
            GPC_HIP(ctx, hipStreamWaitEvent(sc_main, ctx->ev[2][15], 0));
        }
    }
    GPC_HIP(ctx, hipStreamSynchronize(sc_main));
    return GPC_OK;
}

int gpc_dense_fit_predict(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off,
                          const double* x0, const double* x1, const double* y, int ny,
                          int m, const double* xs0, const double* xs1,
                          double* f_star, double* v_star, double* alpha_out, int32_t* status)
{
    return dense_host(ctx, params, P, off, x0, x1, y, ny, m, xs0, xs1, 0.0, 0, false, f_star, v_star, alpha_out, status);
}

int gpc_dense_fit_predict_grid(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off,
                               const double* x0, const double* x1, const double* y, int ny,
                               double res, int sz, double* f_star, double* alpha_out, int32_t* status)
{
    if (ctx && (sz < 0 || sz > 1024)) return gpc_fail(ctx, GPC_EINVAL, "sz out of range");
    return dense_host(ctx, params, P, off, x0, x1, y, ny, sz * sz, nullptr, nullptr, res, sz, true, f_star, nullptr,
                      alpha_out, status);
}

// ------------------------------------------------------------------------------------------------ probit / IRLS (config 5)

void gpc_default_params_irls(gpc_irls_params* p)
{
    if (!p) return;
    p->max_iter = 20;
    p->reserved = 0;
    p->tol = 1e-9;
    p->f_init = 0.0;
}

int gpc_dense_irls_fit_predict_dev(gpc_ctx* ctx, const gpc_params* params, const gpc_irls_params* irls, int P, const int32_t* off,
                                   int n_max, int n_total, const double* x0, const double* x1, const double* y, int m,
                                   const double* xs0, const double* xs1, double res, int sz, double* f_star, double* alpha_out,
                                   double* fhat_out, int32_t* iters, int32_t* status)
{
    const bool grid = (xs0 == nullptr);
    if (ctx && grid && (sz < 0 || sz > 1024)) return gpc_fail(ctx, GPC_EINVAL, "sz out of range");
    if (grid) m = sz * sz;
    int rc = dense_check(ctx, params, P, off, n_max, n_total, x0, x1, y, 1, m, f_star);
    if (rc != GPC_OK) return rc;
    if (!irls) return gpc_fail(ctx, GPC_EINVAL, "irls is NULL");
    if (params->noise_model != 1 && params->noise_model != 2)
        return gpc_fail(ctx, GPC_EINVAL, "the IRLS loop needs a probit noise_model (1 or 2), got %d", params->noise_model);
    if (irls->max_iter < 1 || !(irls->tol >= 0.0) || !(irls->f_init == irls->f_init))
        return gpc_fail(ctx, GPC_EINVAL, "irls parameters out of range");
    if (!(params->noise > 0.0)) return gpc_fail(ctx, GPC_EINVAL, "s20 (params->noise) must be positive");
    if (!grid && m > 0 && !xs1) return gpc_fail(ctx, GPC_EINVAL, "xs1 is NULL");
    if (P == 0) return GPC_OK;
    DenseArgs a{};
    a.prm = *params;
    a.prm.want_variance = 0;
    a.P = P; a.n_max = n_max < 1 ? 1 : n_max; a.n_total = n_total; a.ny = 1; a.m = m;
    a.off = off; a.x0 = x0; a.x1 = x1; a.y = y; a.xs0 = xs0; a.xs1 = xs1;
    a.grid_res = grid ? res : 0.0; a.grid_sz = grid ? sz : 0;
    a.f_star = f_star; a.v_star = nullptr; a.alpha_out = alpha_out; a.status = status;
    IrlsArgs ir{irls->max_iter, irls->tol, irls->f_init, iters, fhat_out};
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    int grid_b = 0;
    const size_t bytes = dense_big_ws_bytes(ctx, a, &grid_b);
    rc = gpc_ws_reserve(ctx, bytes);
    if (rc != GPC_OK) return rc;
    return dense_irls_launch(ctx, a, ir, grid_b);
}

int gpc_dense_irls_fit_predict(gpc_ctx* ctx, const gpc_params* params, const gpc_irls_params* irls, int P, const int32_t* off,
                               const double* x0, const double* x1, const double* y, int m, const double* xs0, const double* xs1,
                               double res, int sz, double* f_star, double* alpha_out, double* fhat_out, int32_t* iters,
                               int32_t* status)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (P < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    const bool grid = (xs0 == nullptr);
    if (grid) {
        if (sz < 0 || sz > 1024) return gpc_fail(ctx, GPC_EINVAL, "sz out of range");
        m = sz * sz;
    }
    int n_max = 0, n_total = 0;
    int rc = host_n_max(ctx, P, off, &n_max, &n_total);
    if (rc != GPC_OK) return rc;
    rc = dense_check(ctx, params, P, off, n_max, n_total, x0, x1, y, 1, m, f_star);
    if (rc != GPC_OK) return rc;
    if (P == 0) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d_off, d_x0, d_x1, d_y, d_xs0, d_xs1, d_f, d_al, d_fh, d_it, d_st;
    const size_t N = (size_t)n_total;
    if ((rc = d_off.alloc(ctx, sizeof(int32_t) * (P + 1))) || (rc = d_x0.alloc(ctx, 8 * N)) || (rc = d_x1.alloc(ctx, 8 * N)) ||
        (rc = d_y.alloc(ctx, 8 * N)) || (rc = d_f.alloc(ctx, 8 * (size_t)P * m)) || (rc = d_al.alloc(ctx, 8 * N)) ||
        (rc = d_fh.alloc(ctx, 8 * N)) || (rc = d_it.alloc(ctx, sizeof(int32_t) * P)) || (rc = d_st.alloc(ctx, sizeof(int32_t) * P)))
        return rc;
    if (!grid && ((rc = d_xs0.alloc(ctx, 8 * (size_t)m)) || (rc = d_xs1.alloc(ctx, 8 * (size_t)m)))) return rc;
    hipStream_t s = gpc_stream_of(ctx);
    GPC_HIP(ctx, hipMemcpyAsync(d_off.p, off, sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice, s));
    if (N) {
        GPC_HIP(ctx, hipMemcpyAsync(d_x0.p, x0, 8 * N, hipMemcpyHostToDevice, s));
        GPC_HIP(ctx, hipMemcpyAsync(d_x1.p, x1, 8 * N, hipMemcpyHostToDevice, s));
        GPC_HIP(ctx, hipMemcpyAsync(d_y.p, y, 8 * N, hipMemcpyHostToDevice, s));
    }
    if (!grid && m) {
        if (!xs1) return gpc_fail(ctx, GPC_EINVAL, "xs1 is NULL");
        GPC_HIP(ctx, hipMemcpyAsync(d_xs0.p, xs0, 8 * (size_t)m, hipMemcpyHostToDevice, s));
        GPC_HIP(ctx, hipMemcpyAsync(d_xs1.p, xs1, 8 * (size_t)m, hipMemcpyHostToDevice, s));
    }
    rc = gpc_dense_irls_fit_predict_dev(ctx, params, irls, P, d_off.as<int32_t>(), n_max, n_total, d_x0.as<double>(), d_x1.as<double>(),
                                        d_y.as<double>(), m, grid ? nullptr : d_xs0.as<double>(), grid ? nullptr : d_xs1.as<double>(),
                                        res, sz, d_f.as<double>(), d_al.as<double>(), d_fh.as<double>(), d_it.as<int32_t>(),
                                        d_st.as<int32_t>());
    if (rc != GPC_OK) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    if (m) GPC_HIP(ctx, hipMemcpyAsync(f_star, d_f.p, 8 * (size_t)P * m, hipMemcpyDeviceToHost, s));
    if (alpha_out && N) GPC_HIP(ctx, hipMemcpyAsync(alpha_out, d_al.p, 8 * N, hipMemcpyDeviceToHost, s));
    if (fhat_out && N) GPC_HIP(ctx, hipMemcpyAsync(fhat_out, d_fh.p, 8 * N, hipMemcpyDeviceToHost, s));
    if (iters) GPC_HIP(ctx, hipMemcpyAsync(iters, d_it.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost, s));
    if (status) GPC_HIP(ctx, hipMemcpyAsync(status, d_st.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

__global__ void gpc_noise_eval_kernel(int model, double s20, int n, const double* y, const double* x, const double* sx, double* q, double* r)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qq, rr;
    if (model == GPC_NOISE_GAUSSIAN) gpc_gaussian_q_r(s20, y[i], x[i], sx[i], &qq, &rr);
    else gpc_probit_q_r(model, s20, y[i], x[i], sx[i], &qq, &rr);
    q[i] = qq;
    r[i] = rr;
}

int gpc_noise_eval(gpc_ctx* ctx, int noise_model, double s20, int n, const double* y, const double* x, const double* sigma_x,
                   double* q, double* r)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (noise_model < 0 || noise_model > 2) return gpc_fail(ctx, GPC_EINVAL, "noise_model must be 0, 1 or 2");
    if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (n == 0) return GPC_OK;
    if (!y || !x || !sigma_x || !q || !r) return gpc_fail(ctx, GPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    const size_t nb = 8 * (size_t)n;
    int rc = d.alloc(ctx, 5 * nb);
    if (rc != GPC_OK) return rc;
    char* b = d.as<char>();
    hipStream_t s = ctx->stream;
    GPC_HIP(ctx, hipMemcpyAsync(b, y, nb, hipMemcpyHostToDevice, s));
    GPC_HIP(ctx, hipMemcpyAsync(b + nb, x, nb, hipMemcpyHostToDevice, s));
    GPC_HIP(ctx, hipMemcpyAsync(b + 2 * nb, sigma_x, nb, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gpc_noise_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, s, noise_model, s20, n, (const double*)b,
                       (const double*)(b + nb), (const double*)(b + 2 * nb), (double*)(b + 3 * nb), (double*)(b + 4 * nb));
    GPC_HIP(ctx, hipGetLastError());
    GPC_HIP(ctx, hipMemcpyAsync(q, b + 3 * nb, nb, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipMemcpyAsync(r, b + 4 * nb, nb, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

// ------------------------------------------------------------------------------------------------ partition

// Patches are independent (/root/reference/src/gp_compressor.cpp:146-163), so one process per GPU takes a subset.
// Longest-processing-time: sort by cost descending, give each to the least-loaded rank that still has a free slot.
int gpc_partition_patches(int P, const int32_t* off, int world, int sparse_capacity, int32_t* slot_patch)
{
    if (P < 0 || world <= 0 || !slot_patch || (P > 0 && !off)) return GPC_EINVAL;
    const int S = (P + world - 1) / world;
    for (long i = 0; i < (long)S * world; ++i) slot_patch[i] = -1;
    std::vector<int> order(P);
    std::iota(order.begin(), order.end(), 0);
    auto cost = [&](int p) -> double {
        double n = (double)(off[p + 1] - off[p]);
        if (sparse_capacity > 0) {
            double b = std::min(n, (double)sparse_capacity);
            return n * b * b;
        }
        return n * n * n;
    };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
    std::vector<double> load(world, 0.0);
    std::vector<int> used(world, 0);
    for (int p : order) {
        int best = -1;
        for (int r = 0; r < world; ++r)
            if (used[r] < S && (best < 0 || load[r] < load[best])) best = r;
        slot_patch[(long)best * S + used[best]] = p;
        used[best]++;
        load[best] += cost(p);
    }
    return GPC_OK;
}

// ------------------------------------------------------------------------------------------------ test hooks
// Host-side evaluation of the device exp() (same source, gpc_device.h) so that the CPU suite can bound its error.
void gpc_test_exp_host(const double* x, double* out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = gpc_exp_tbl(x[i], h_gpc_exp_table);
}
void gpc_test_exp_small_host(const double* x, double* out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = gpc_exp_small(x[i]);
}

}  // extern "C"
