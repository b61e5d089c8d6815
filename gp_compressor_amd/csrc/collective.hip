// collective.hip -- the ONE exchange step of the multi-GPU path: patches are independent
// (/root/reference/src/gp_compressor.cpp:146-163), ranks own S = ceil(P / world) slots each (gpc_partition_patches), and a single
// ncclAllGather of the slot buffers over RCCL / xGMI followed by a device gather (un-permutation to patch order) reassembles
// the decompressed grids on every rank.  RCCL is bound at run time (dlopen): libgpc_hip.so has no link-time dependency on it, a
// process that never creates a gpc_comm never loads it, and a process that already holds a copy (PyTorch ships its own) shares
// that copy -- which is what lets gpc_comm_adopt take an ncclComm_t the caller created.
#include <dlfcn.h>

#include <vector>

#include "gpc_internal.h"

namespace {

struct nccl_unique_id { char internal[128]; };   // ncclUniqueId (rccl.h): NCCL_UNIQUE_ID_BYTES = 128, passed by value
typedef void* nccl_comm_t;
enum { NCCL_FLOAT64 = 8 };                       // ncclDataType_t: ncclFloat64 / ncclDouble

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string where;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;

// An already-loaded copy first (same soname: PyTorch's torch/lib/librccl.so is "librccl.so.1" too), then the system's.
const RcclApi* rccl_api(std::string* err)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return &g_rccl;
    void* h = nullptr;
    std::string tried;
    const char* env = getenv("GPC_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            if (!n || h) continue;
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) g_rccl.where = std::string(n) + (pass == 0 ? " (already loaded in the process)" : "");
            else if (pass == 1) tried += std::string(n) + " ";
        }
    if (!h) {
        if (err) *err = "RCCL not found (tried " + tried + "; set GPC_RCCL_PATH)";
        return nullptr;
    }
    RcclApi a;
    a.lib = h;
    a.where = g_rccl.where;
#define GPC_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
    GPC_SYM(GetUniqueId, "ncclGetUniqueId");
    GPC_SYM(CommInitRank, "ncclCommInitRank");
    GPC_SYM(CommInitAll, "ncclCommInitAll");
    GPC_SYM(CommDestroy, "ncclCommDestroy");
    GPC_SYM(AllGather, "ncclAllGather");
    GPC_SYM(GroupStart, "ncclGroupStart");
    GPC_SYM(GroupEnd, "ncclGroupEnd");
    GPC_SYM(GetErrorString, "ncclGetErrorString");
#undef GPC_SYM
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd) {
        if (err) *err = "the RCCL library lacks an expected symbol";
        dlclose(h);
        return nullptr;
    }
    g_rccl = a;
    return &g_rccl;
}

__global__ void unpermute_rows_kernel(int P, int row_doubles, const int32_t* __restrict__ slot_of_patch,
                                      const double* __restrict__ gathered, double* __restrict__ f_star)
{
    // one workgroup per patch row; 16-byte accesses when the row length allows
    const int p = blockIdx.x;
    if (p >= P) return;
    const double* src = gathered + (size_t)slot_of_patch[p] * row_doubles;
    double* dst = f_star + (size_t)p * row_doubles;
    if ((row_doubles & 1) == 0) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
        double2* d2 = reinterpret_cast<double2*>(dst);
        for (int i = threadIdx.x; i < row_doubles / 2; i += blockDim.x) d2[i] = s2[i];
    } else {
        for (int i = threadIdx.x; i < row_doubles; i += blockDim.x) dst[i] = src[i];
    }
}

}  // namespace

struct gpc_comm {
    gpc_ctx* ctx = nullptr;
    nccl_comm_t comm = nullptr;
    bool owned = true;
    int world = 1, rank = 0;
    int P = 0, S = 0;                      // partition: P patches in world * S slots
    int32_t* d_slot_of_patch = nullptr;    // [P] row of the gathered buffer that holds patch p
};

extern "C" {

int gpc_comm_unique_id(void* id128)
{
    if (!id128) return GPC_EINVAL;
    std::string err;
    const RcclApi* R = rccl_api(&err);
    if (!R) return GPC_ENODEV;
    nccl_unique_id id;
    if (R->GetUniqueId(&id) != 0) return GPC_EHIP;
    std::memcpy(id128, &id, sizeof(id));
    return GPC_OK;
}

static int comm_wrap(gpc_ctx* ctx, nccl_comm_t c, bool owned, int world, int rank, gpc_comm** out)
{
    gpc_comm* g = new (std::nothrow) gpc_comm();
    if (!g) return GPC_ENOMEM;
    g->ctx = ctx; g->comm = c; g->owned = owned; g->world = world; g->rank = rank;
    gpc_ctx_ref(ctx);
    *out = g;
    return GPC_OK;
}

int gpc_comm_create(gpc_ctx* ctx, int world, int rank, const void* id128, gpc_comm** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !id128) return gpc_fail(ctx, GPC_EINVAL, "bad world / rank / id");
    std::string err;
    const RcclApi* R = rccl_api(&err);
    if (!R) return gpc_fail(ctx, GPC_ENODEV, "%s", err.c_str());
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    nccl_unique_id id;
    std::memcpy(&id, id128, sizeof(id));
    nccl_comm_t c = nullptr;
    const int rc = R->CommInitRank(&c, world, id, rank);
    if (rc != 0) return gpc_fail(ctx, GPC_EHIP, "ncclCommInitRank: %s", R->GetErrorString ? R->GetErrorString(rc) : "error");
    return comm_wrap(ctx, c, true, world, rank, out);
}

int gpc_comm_create_all(int ndev, gpc_ctx* const* ctxs, gpc_comm** out)
{
    if (ndev < 1 || !ctxs || !out) return GPC_EINVAL;
    for (int i = 0; i < ndev; ++i) {
        out[i] = nullptr;
        if (!ctxs[i] || ctxs[i]->dead.load()) return GPC_EINVAL;
    }
    std::string err;
    const RcclApi* R = rccl_api(&err);
    if (!R) return gpc_fail(ctxs[0], GPC_ENODEV, "%s", err.c_str());
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) devs[i] = ctxs[i]->device;
    std::vector<nccl_comm_t> cs(ndev, nullptr);
    const int rc = R->CommInitAll(cs.data(), ndev, devs.data());
    if (rc != 0) return gpc_fail(ctxs[0], GPC_EHIP, "ncclCommInitAll: %s", R->GetErrorString ? R->GetErrorString(rc) : "error");
    for (int i = 0; i < ndev; ++i) {
        const int r2 = comm_wrap(ctxs[i], cs[i], true, ndev, i, &out[i]);
        if (r2 != GPC_OK) {
            // all or nothing: the wrapped ones go (with their context references), the raw communicators not yet wrapped too
            for (int j = 0; j < i; ++j) { gpc_comm_destroy(out[j]); out[j] = nullptr; }
            for (int j = i; j < ndev; ++j)
                if (cs[j] && R->CommDestroy) (void)R->CommDestroy(cs[j]);
            return r2;
        }
    }
    return GPC_OK;
}

int gpc_comm_adopt(gpc_ctx* ctx, void* nccl_comm, int world, int rank, gpc_comm** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) return gpc_fail(ctx, GPC_EINVAL, "bad communicator / world / rank");
    std::string err;
    if (!rccl_api(&err)) return gpc_fail(ctx, GPC_ENODEV, "%s", err.c_str());
    return comm_wrap(ctx, nccl_comm, false, world, rank, out);
}

void gpc_comm_destroy(gpc_comm* c)
{
    if (!c) return;
    gpc_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    if (!ctx->dead.load()) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->dead.load()) (void)hipStreamSynchronize(ctx->stream);
    }
    if (c->d_slot_of_patch) (void)hipFree(c->d_slot_of_patch);
    if (c->owned && c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    gpc_ctx_unref(ctx);
}

int gpc_comm_world(const gpc_comm* c) { return c ? c->world : GPC_EINVAL; }
int gpc_comm_rank(const gpc_comm* c) { return c ? c->rank : GPC_EINVAL; }
const char* gpc_comm_library(void) { return g_rccl.lib ? g_rccl.where.c_str() : ""; }

int gpc_comm_set_partition(gpc_comm* c, int P, const int32_t* slot_patch)
{
    if (!c) return GPC_EINVAL;
    gpc_ctx* ctx = c->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;
    if (P < 0 || (P > 0 && !slot_patch)) return gpc_fail(ctx, GPC_EINVAL, "bad partition");
    const int S = (P + c->world - 1) / c->world;
    std::vector<int32_t> inv((size_t)(P > 0 ? P : 1), -1);
    for (long s = 0; s < (long)S * c->world; ++s) {
        const int p = slot_patch[s];
        if (p < -1 || p >= P) return gpc_fail(ctx, GPC_EINVAL, "slot %ld holds patch %d of %d", s, p, P);
        if (p >= 0) {
            if (inv[p] != -1) return gpc_fail(ctx, GPC_EINVAL, "patch %d sits in two slots", p);
            inv[p] = (int32_t)s;
        }
    }
    for (int p = 0; p < P; ++p)
        if (inv[p] < 0) return gpc_fail(ctx, GPC_EINVAL, "patch %d has no slot", p);
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    GPC_HIP(ctx, hipStreamSynchronize(ctx->stream));             // an un-permutation in flight may still read the old table
    if (c->d_slot_of_patch) { (void)hipFree(c->d_slot_of_patch); c->d_slot_of_patch = nullptr; }
    GPC_HIP(ctx, hipMalloc(&c->d_slot_of_patch, sizeof(int32_t) * inv.size()));
    GPC_HIP(ctx, hipMemcpy(c->d_slot_of_patch, inv.data(), sizeof(int32_t) * inv.size(), hipMemcpyHostToDevice));
    c->P = P;
    c->S = S;
    return GPC_OK;
}

int gpc_group_start(void)
{
    const RcclApi* R = rccl_api(nullptr);
    return R ? (R->GroupStart() == 0 ? GPC_OK : GPC_EHIP) : GPC_ENODEV;
}
int gpc_group_end(void)
{
    const RcclApi* R = rccl_api(nullptr);
    return R ? (R->GroupEnd() == 0 ? GPC_OK : GPC_EHIP) : GPC_ENODEV;
}

int gpc_allgather_fstar_dev(gpc_comm* c, int row_doubles, const double* local_f, double* gathered, double* f_star)
{
    if (!c) return GPC_EINVAL;
    gpc_ctx* ctx = c->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;
    if (row_doubles < 1) return gpc_fail(ctx, GPC_EINVAL, "row_doubles must be positive");
    if (!c->d_slot_of_patch) return gpc_fail(ctx, GPC_EINVAL, "gpc_comm_set_partition has not been called");
    if (c->P == 0) return GPC_OK;
    if (!local_f || !gathered) return gpc_fail(ctx, GPC_EINVAL, "local_f / gathered is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = g_rccl.AllGather(local_f, gathered, (size_t)c->S * (size_t)row_doubles, NCCL_FLOAT64, c->comm, ctx->stream);
    if (rc != 0) return gpc_fail(ctx, GPC_EHIP, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    if (f_star) {
        hipLaunchKernelGGL(unpermute_rows_kernel, dim3(c->P), dim3(128), 0, ctx->stream, c->P, row_doubles, c->d_slot_of_patch, gathered, f_star);
        GPC_HIP(ctx, hipGetLastError());
    }
    return GPC_OK;
}

int gpc_unpermute_fstar_dev(gpc_comm* c, int row_doubles, const double* gathered, double* f_star)
{
    if (!c) return GPC_EINVAL;
    gpc_ctx* ctx = c->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;
    if (row_doubles < 1) return gpc_fail(ctx, GPC_EINVAL, "row_doubles must be positive");
    if (!c->d_slot_of_patch) return gpc_fail(ctx, GPC_EINVAL, "gpc_comm_set_partition has not been called");
    if (c->P == 0) return GPC_OK;
    if (!gathered || !f_star) return gpc_fail(ctx, GPC_EINVAL, "gathered / f_star is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(unpermute_rows_kernel, dim3(c->P), dim3(128), 0, ctx->stream, c->P, row_doubles, c->d_slot_of_patch, gathered, f_star);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

}  // extern "C"
