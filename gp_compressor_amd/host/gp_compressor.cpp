// gp_compressor.cpp -- host-side mirror of /root/reference/src/gp_compressor.cpp on top of the C-ABI (include/gpc.h).
#include "gp_compressor.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <tuple>
#include <unordered_map>

namespace gpc {

double rbf_kernel::kernel_function(const double xi[2], const double xj[2]) const
{
    const double d0 = xi[0] - xj[0], d1 = xi[1] - xj[1];
    return p_[0] * std::exp(-0.5f / p_[1] * (d0 * d0 + d1 * d1));   // src/rbf_kernel.cpp:17
}

namespace {

struct VoxelKey {
    int x, y, z;
    bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator<(const VoxelKey& o) const { return std::tie(z, y, x) < std::tie(o.z, o.y, o.x); }
};
struct VoxelHash {
    size_t operator()(const VoxelKey& k) const
    {
        return (size_t)k.x * 73856093u ^ (size_t)k.y * 19349663u ^ (size_t)k.z * 83492791u;
    }
};

// eigenvector of the smallest eigenvalue of a symmetric 4x4 matrix (cyclic Jacobi): the last right singular vector of
// the k x 4 homogeneous point matrix, which is what JacobiSVD(...).matrixV().col(3) gives (src/gp_compressor.cpp:35-36)
void smallest_eigvec4(double A[4][4], double v[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double offd = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) offd += A[p][q] * A[p][q];
        if (offd == 0.0) break;
        for (int p = 0; p < 4; ++p) {
            for (int q = p + 1; q < 4; ++q) {
                if (A[p][q] == 0.0) continue;
                const double g = 100.0 * std::fabs(A[p][q]);     // negligible against both diagonal entries: drop it
                if (std::fabs(A[p][p]) + g == std::fabs(A[p][p]) && std::fabs(A[q][q]) + g == std::fabs(A[q][q])) {
                    A[p][q] = A[q][p] = 0.0;
                    continue;
                }
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                A[p][q] = A[q][p] = 0.0;                       // the rotation annihilates this pair: make it exact
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (A[i][i] < A[best][best]) best = i;
    for (int k = 0; k < 4; ++k) v[k] = V[k][best];
}

void cross(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
void normalize(double a[3])
{
    const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n > 0) { a[0] /= n; a[1] /= n; a[2] /= n; }
}

void check(int rc, gpc_ctx* ctx, const char* what)
{
    if (rc != GPC_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? gpc_last_error(ctx) : "gpc error") +
                                               " (code " + std::to_string(rc) + ")");
}

}  // namespace

gp_compressor::gp_compressor(const pointcloud& ncloud, double res, int sz, gp_model model, int device)
    : rng([] { return std::rand(); }), cloud_(ncloud), res_(res), sz_(sz), model_(model), device_(device)
{
    gpc_default_params_sparse(&depth_params, 1);
    gpc_default_params_sparse(&rgb_params, 3);
    gpc_default_params_dense(&dense_params);
}

void gp_compressor::release_device()
{
    if (dev_patches_) { gpc_patches_destroy(dev_patches_); dev_patches_ = nullptr; }
    if (ctx_) {
        if (d_dense_f_) (void)gpc_dev_free(ctx_, d_dense_f_);
        if (d_dense_c_) (void)gpc_dev_free(ctx_, d_dense_c_);
    }
    d_dense_f_ = d_dense_c_ = nullptr;
}

void gp_compressor::set_devices(const std::vector<int>& devices)
{
    for (gpc_sparse* q : shard_gps_) if (q) gpc_sparse_destroy(q);
    for (gpc_sparse* q : shard_rgb_) if (q) gpc_sparse_destroy(q);
    shard_gps_.clear();
    shard_rgb_.clear();
    shard_slots_.clear();
    for (gpc_comm* c : shard_comm_) gpc_comm_destroy(c);
    for (gpc_ctx* c : shard_ctx_) gpc_ctx_destroy(c);
    shard_comm_.clear();
    shard_ctx_.clear();
    devices_ = devices;
}

// one context per device and the communicators of the one exchange
void gp_compressor::init_shards()
{
    if (!shard_ctx_.empty()) return;
    const int world = (int)devices_.size();
    // built into locals and committed to the members only when every context and communicator exists: a failure half-way
    // leaves the object in its single-device state, never with null handles that a later call would pass on
    std::vector<gpc_ctx*> ctxs(world, nullptr);
    std::vector<gpc_comm*> comms(world, nullptr);
    try {
        for (int r = 0; r < world; ++r) check(gpc_ctx_create(&ctxs[r], devices_[r]), nullptr, "gpc_ctx_create (shard)");
        check(gpc_comm_create_all(world, ctxs.data(), comms.data()), ctxs[0], "gpc_comm_create_all");
    } catch (...) {
        for (gpc_comm* c : comms) if (c) gpc_comm_destroy(c);
        for (gpc_ctx* c : ctxs) if (c) gpc_ctx_destroy(c);
        throw;
    }
    shard_ctx_.swap(ctxs);
    shard_comm_.swap(comms);
}

// The sparse model over several GPUs (BASELINE configs[3]; the reference keeps adding to trained GPs,
// /root/reference/src/gp_mapping.cpp:338-339): gpc_partition_patches with the sparse cost model deals the patches ONCE, the
// depth and colour GPs of a device's slots are created there and stay there.  perm_d / perm_c: the insertion orders in
// patch order, as train_processes() drew them.
void gp_compressor::train_sparse_sharded(const std::vector<int32_t>& perm_d, const std::vector<int32_t>& perm_c)
{
    const int P = batch_.patches(), world = (int)devices_.size();
    init_shards();
    const int S = (P + world - 1) / world;
    if (shard_slots_.empty()) {
        // built in locals and committed to the members only when every create has succeeded (as init_shards() does): a create that
        // throws midway must not leave shard_slots_ filled beside half-created GPs
        std::vector<int32_t> slots((size_t)S * world, -1);
        check(gpc_partition_patches(P, batch_.off.data(), world, depth_params.capacity > 0 ? depth_params.capacity : 1, slots.data()),
              nullptr, "gpc_partition_patches");
        std::vector<gpc_sparse*> gps(world, nullptr), rgb(world, nullptr);
        try {
            for (int r = 0; r < world; ++r) {
                check(gpc_sparse_create(shard_ctx_[r], &depth_params, S, 1, &gps[r]), shard_ctx_[r], "gpc_sparse_create(depth, shard)");
                check(gpc_sparse_create(shard_ctx_[r], &rgb_params, S, 3, &rgb[r]), shard_ctx_[r], "gpc_sparse_create(rgb, shard)");
            }
        } catch (...) {
            for (int r = 0; r < world; ++r) {
                if (gps[r]) gpc_sparse_destroy(gps[r]);
                if (rgb[r]) gpc_sparse_destroy(rgb[r]);
            }
            throw;
        }
        shard_slots_.swap(slots);
        shard_gps_.swap(gps);
        shard_rgb_.swap(rgb);
    }
    const size_t N = batch_.x0.size();
    struct Shard {
        std::vector<int32_t> off, pd, pc;
        std::vector<double> x0, x1, y, rgb;
        int n_max = 0;
        void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_rgb = nullptr, *d_pd = nullptr, *d_pc = nullptr, *d_st = nullptr;
    };
    std::vector<Shard> sh(world);
    auto free_all = [&]() {
        for (int r = 0; r < world; ++r) {
            (void)gpc_ctx_synchronize(shard_ctx_[r]);
            for (void* q : {sh[r].d_off, sh[r].d_x0, sh[r].d_x1, sh[r].d_y, sh[r].d_rgb, sh[r].d_pd, sh[r].d_pc, sh[r].d_st})
                if (q) (void)gpc_dev_free(shard_ctx_[r], q);
        }
    };
    try {
        for (int r = 0; r < world; ++r) {
            Shard& s = sh[r];
            gpc_ctx* c = shard_ctx_[r];
            s.off.assign(S + 1, 0);
            for (int q = 0; q < S; ++q) {
                const int p = shard_slots_[(size_t)r * S + q];
                const int n = p >= 0 ? batch_.off[p + 1] - batch_.off[p] : 0;
                s.off[q + 1] = s.off[q] + n;
                s.n_max = std::max(s.n_max, n);
            }
            const size_t Nl = (size_t)s.off[S];
            s.x0.resize(Nl); s.x1.resize(Nl); s.y.resize(Nl); s.rgb.resize(3 * Nl); s.pd.resize(Nl); s.pc.resize(Nl);
            for (int q = 0; q < S; ++q) {
                const int p = shard_slots_[(size_t)r * S + q];
                if (p < 0) continue;
                const size_t o = (size_t)batch_.off[p], n = (size_t)(batch_.off[p + 1] - batch_.off[p]), lo = (size_t)s.off[q];
                std::copy_n(&batch_.x0[o], n, &s.x0[lo]);
                std::copy_n(&batch_.x1[o], n, &s.x1[lo]);
                std::copy_n(&batch_.y[o], n, &s.y[lo]);
                std::copy_n(&perm_d[o], n, &s.pd[lo]);            // patch-local indices: they travel with the patch
                std::copy_n(&perm_c[o], n, &s.pc[lo]);
                for (int a = 0; a < 3; ++a) std::copy_n(&batch_.rgb[a * N + o], n, &s.rgb[a * Nl + lo]);
            }
            auto up = [&](void** d, const void* h, size_t bytes) {
                check(gpc_dev_malloc(c, bytes ? bytes : 8, d), c, "gpc_dev_malloc (shard)");
                if (bytes) check(gpc_dev_memcpy(c, *d, h, bytes, GPC_COPY_H2D), c, "gpc_dev_memcpy (shard)");
            };
            up(&s.d_off, s.off.data(), sizeof(int32_t) * (size_t)(S + 1));
            up(&s.d_x0, s.x0.data(), 8 * Nl);
            up(&s.d_x1, s.x1.data(), 8 * Nl);
            up(&s.d_y, s.y.data(), 8 * Nl);
            up(&s.d_rgb, s.rgb.data(), 24 * Nl);
            up(&s.d_pd, s.pd.data(), 4 * Nl);
            up(&s.d_pc, s.pc.data(), 4 * Nl);
            check(gpc_dev_malloc(c, sizeof(int32_t) * (size_t)S, &s.d_st), c, "gpc_dev_malloc");
        }
        // the add calls only enqueue: the devices run side by side
        for (int r = 0; r < world; ++r) {
            Shard& s = sh[r];
            check(gpc_sparse_add_dev(shard_gps_[r], (const int32_t*)s.d_off, s.n_max, s.off[S], (const double*)s.d_x0, (const double*)s.d_x1,
                                     (const double*)s.d_y, (const int32_t*)s.d_pd, (int32_t*)s.d_st), shard_ctx_[r], "gpc_sparse_add_dev (shard, depth)");
            check(gpc_sparse_add_dev(shard_rgb_[r], (const int32_t*)s.d_off, s.n_max, s.off[S], (const double*)s.d_x0, (const double*)s.d_x1,
                                     (const double*)s.d_rgb, (const int32_t*)s.d_pc, (int32_t*)s.d_st), shard_ctx_[r], "gpc_sparse_add_dev (shard, rgb)");
        }
        std::vector<int32_t> st(S);
        for (int r = 0; r < world; ++r) {
            check(gpc_dev_memcpy(shard_ctx_[r], st.data(), sh[r].d_st, sizeof(int32_t) * (size_t)S, GPC_COPY_D2H), shard_ctx_[r], "download status");
            for (int q = 0; q < S; ++q) {
                const int p = shard_slots_[(size_t)r * S + q];
                if (p >= 0) status_[p] = st[q];
            }
        }
    } catch (...) {
        free_all();
        throw;
    }
    free_all();
}

// load_compressed() of the sharded sparse model: every device predicts the grids of its slots, ONE grouped all-gather of the depth and
// the colour grids reassembles them on device 0 in patch order; basis sizes come back slot by slot.
void gp_compressor::predict_sparse_sharded(const std::vector<double>& xs0, const std::vector<double>& xs1, std::vector<double>& f_star,
                                           std::vector<double>& c_star, std::vector<int32_t>& bv)
{
    const int P = batch_.patches(), m = sz_ * sz_, world = (int)devices_.size();
    const int S = (P + world - 1) / world;
    struct Buf { void *xs0 = nullptr, *xs1 = nullptr, *lf = nullptr, *lc = nullptr, *gf = nullptr, *gc = nullptr, *f = nullptr, *c = nullptr; };
    std::vector<Buf> b(world);
    bool in_group = false;
    auto free_all = [&]() {
        for (int r = 0; r < world; ++r) {
            (void)gpc_ctx_synchronize(shard_ctx_[r]);
            for (void* q : {b[r].xs0, b[r].xs1, b[r].lf, b[r].lc, b[r].gf, b[r].gc, b[r].f, b[r].c})
                if (q) (void)gpc_dev_free(shard_ctx_[r], q);
        }
    };
    try {
        for (int r = 0; r < world; ++r) {
            gpc_ctx* c = shard_ctx_[r];
            check(gpc_dev_malloc(c, 8 * (size_t)m, &b[r].xs0), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)m, &b[r].xs1), c, "gpc_dev_malloc");
            check(gpc_dev_memcpy(c, b[r].xs0, xs0.data(), 8 * (size_t)m, GPC_COPY_H2D), c, "upload grid");
            check(gpc_dev_memcpy(c, b[r].xs1, xs1.data(), 8 * (size_t)m, GPC_COPY_H2D), c, "upload grid");
            check(gpc_dev_malloc(c, 8 * (size_t)S * m, &b[r].lf), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * 3 * m, &b[r].lc), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * world * m, &b[r].gf), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * world * 3 * m, &b[r].gc), c, "gpc_dev_malloc");
            if (r == 0) {
                check(gpc_dev_malloc(c, 8 * (size_t)P * m, &b[r].f), c, "gpc_dev_malloc");
                check(gpc_dev_malloc(c, 8 * (size_t)P * 3 * m, &b[r].c), c, "gpc_dev_malloc");
            }
            check(gpc_comm_set_partition(shard_comm_[r], P, shard_slots_.data()), c, "gpc_comm_set_partition");
        }
        for (int r = 0; r < world; ++r) {
            check(gpc_sparse_predict_dev(shard_gps_[r], m, (const double*)b[r].xs0, (const double*)b[r].xs1, (double*)b[r].lf, nullptr, 0, nullptr),
                  shard_ctx_[r], "gpc_sparse_predict_dev (shard, depth)");
            check(gpc_sparse_predict_dev(shard_rgb_[r], m, (const double*)b[r].xs0, (const double*)b[r].xs1, (double*)b[r].lc, nullptr, 0, nullptr),
                  shard_ctx_[r], "gpc_sparse_predict_dev (shard, rgb)");
        }
        check(gpc_group_start(), shard_ctx_[0], "gpc_group_start");
        in_group = true;
        for (int r = 0; r < world; ++r) {
            check(gpc_allgather_fstar_dev(shard_comm_[r], m, (const double*)b[r].lf, (double*)b[r].gf, nullptr), shard_ctx_[r], "all-gather f*");
            check(gpc_allgather_fstar_dev(shard_comm_[r], 3 * m, (const double*)b[r].lc, (double*)b[r].gc, nullptr), shard_ctx_[r], "all-gather c*");
        }
        in_group = false;
        check(gpc_group_end(), shard_ctx_[0], "gpc_group_end");
        check(gpc_unpermute_fstar_dev(shard_comm_[0], m, (const double*)b[0].gf, (double*)b[0].f), shard_ctx_[0], "un-permute f*");
        check(gpc_unpermute_fstar_dev(shard_comm_[0], 3 * m, (const double*)b[0].gc, (double*)b[0].c), shard_ctx_[0], "un-permute c*");
        f_star.assign((size_t)P * m, 0.0);
        c_star.assign((size_t)P * 3 * m, 0.0);
        check(gpc_dev_memcpy(shard_ctx_[0], f_star.data(), b[0].f, 8 * (size_t)P * m, GPC_COPY_D2H), shard_ctx_[0], "download f*");
        check(gpc_dev_memcpy(shard_ctx_[0], c_star.data(), b[0].c, 8 * (size_t)P * 3 * m, GPC_COPY_D2H), shard_ctx_[0], "download c*");
        std::vector<int32_t> sz_slot(S);
        bv.assign(P, 0);
        for (int r = 0; r < world; ++r) {
            check(gpc_sparse_sizes(shard_gps_[r], sz_slot.data()), shard_ctx_[r], "gpc_sparse_sizes (shard)");
            for (int q = 0; q < S; ++q) {
                const int p = shard_slots_[(size_t)r * S + q];
                if (p >= 0) bv[p] = sz_slot[q];
            }
        }
    } catch (...) {
        if (in_group) (void)gpc_group_end();
        free_all();
        throw;
    }
    free_all();
}

// The dense model over several GPUs: the batched form of "patches shard embarrassingly, one all-gather reassembles the
// decompressed cloud" (BASELINE north_star; patches are independent, src/gp_compressor.cpp:146-163).
void gp_compressor::train_dense_sharded()
{
    const int P = batch_.patches(), m = sz_ * sz_, world = (int)devices_.size();
    init_shards();
    const int S = (P + world - 1) / world;
    std::vector<int32_t> slots((size_t)S * world);
    check(gpc_partition_patches(P, batch_.off.data(), world, 0, slots.data()), nullptr, "gpc_partition_patches");
    const size_t N = batch_.x0.size();
    struct Shard {
        std::vector<int32_t> off;
        std::vector<double> x0, x1, y, rgb;
        int n_max = 0;
        void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_rgb = nullptr, *d_st = nullptr;
        void *d_lf = nullptr, *d_lc = nullptr, *d_gf = nullptr, *d_gc = nullptr, *d_f = nullptr, *d_c = nullptr;
    };
    std::vector<Shard> sh(world);
    bool in_group = false;                    // an RCCL group left open would swallow every later collective of the process
    auto free_all = [&]() {
        for (int r = 0; r < world; ++r)
            for (void* q : {sh[r].d_off, sh[r].d_x0, sh[r].d_x1, sh[r].d_y, sh[r].d_rgb, sh[r].d_st, sh[r].d_lf, sh[r].d_lc, sh[r].d_gf,
                            sh[r].d_gc, sh[r].d_f, sh[r].d_c})
                if (q) (void)gpc_dev_free(shard_ctx_[r], q);
    };
    try {
        for (int r = 0; r < world; ++r) {
            Shard& s = sh[r];
            gpc_ctx* c = shard_ctx_[r];
            // this device's slots as a CSR batch (padding slots are empty patches)
            s.off.assign(S + 1, 0);
            for (int q = 0; q < S; ++q) {
                const int p = slots[(size_t)r * S + q];
                const int n = p >= 0 ? batch_.off[p + 1] - batch_.off[p] : 0;
                s.off[q + 1] = s.off[q] + n;
                s.n_max = std::max(s.n_max, n);
            }
            const size_t Nl = (size_t)s.off[S];
            s.x0.resize(Nl); s.x1.resize(Nl); s.y.resize(Nl); s.rgb.resize(3 * Nl);
            for (int q = 0; q < S; ++q) {
                const int p = slots[(size_t)r * S + q];
                if (p < 0) continue;
                const size_t o = (size_t)batch_.off[p], n = (size_t)(batch_.off[p + 1] - batch_.off[p]), lo = (size_t)s.off[q];
                std::copy_n(&batch_.x0[o], n, &s.x0[lo]);
                std::copy_n(&batch_.x1[o], n, &s.x1[lo]);
                std::copy_n(&batch_.y[o], n, &s.y[lo]);
                for (int a = 0; a < 3; ++a) std::copy_n(&batch_.rgb[a * N + o], n, &s.rgb[a * Nl + lo]);
            }
            auto up = [&](void** d, const void* h, size_t bytes) {
                check(gpc_dev_malloc(c, bytes, d), c, "gpc_dev_malloc (shard)");
                if (bytes) check(gpc_dev_memcpy(c, *d, h, bytes, GPC_COPY_H2D), c, "gpc_dev_memcpy (shard)");
            };
            up(&s.d_off, s.off.data(), sizeof(int32_t) * (size_t)(S + 1));
            up(&s.d_x0, s.x0.data(), 8 * Nl);
            up(&s.d_x1, s.x1.data(), 8 * Nl);
            up(&s.d_y, s.y.data(), 8 * Nl);
            up(&s.d_rgb, s.rgb.data(), 24 * Nl);
            check(gpc_dev_malloc(c, sizeof(int32_t) * (size_t)S, &s.d_st), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * m, &s.d_lf), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * 3 * m, &s.d_lc), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * world * m, &s.d_gf), c, "gpc_dev_malloc");
            check(gpc_dev_malloc(c, 8 * (size_t)S * world * 3 * m, &s.d_gc), c, "gpc_dev_malloc");
            if (r == 0) {
                check(gpc_dev_malloc(c, 8 * (size_t)P * m, &s.d_f), c, "gpc_dev_malloc");
                check(gpc_dev_malloc(c, 8 * (size_t)P * 3 * m, &s.d_c), c, "gpc_dev_malloc");
            }
            check(gpc_comm_set_partition(shard_comm_[r], P, slots.data()), c, "gpc_comm_set_partition");
        }
        // every device fits + predicts its slots: the launches only enqueue, so the devices run concurrently
        for (int r = 0; r < world; ++r) {
            Shard& s = sh[r];
            gpc_ctx* c = shard_ctx_[r];
            check(gpc_dense_fit_predict_grid_dev(c, &dense_params, S, (const int32_t*)s.d_off, s.n_max, s.off[S], (const double*)s.d_x0,
                                                 (const double*)s.d_x1, (const double*)s.d_y, 1, res_, sz_, (double*)s.d_lf, nullptr,
                                                 (int32_t*)s.d_st), c, "gpc_dense_fit_predict_grid_dev (shard, depth)");
            check(gpc_dense_fit_predict_grid_dev(c, &dense_params, S, (const int32_t*)s.d_off, s.n_max, s.off[S], (const double*)s.d_x0,
                                                 (const double*)s.d_x1, (const double*)s.d_rgb, 3, res_, sz_, (double*)s.d_lc, nullptr,
                                                 nullptr), c, "gpc_dense_fit_predict_grid_dev (shard, rgb)");
        }
        // the one exchange: all-gather of the depth grids and of the colour grids (inside one bracket: one fused exchange)
        check(gpc_group_start(), shard_ctx_[0], "gpc_group_start");
        in_group = true;
        for (int r = 0; r < world; ++r) {
            check(gpc_allgather_fstar_dev(shard_comm_[r], m, (const double*)sh[r].d_lf, (double*)sh[r].d_gf, nullptr), shard_ctx_[r], "all-gather f*");
            check(gpc_allgather_fstar_dev(shard_comm_[r], 3 * m, (const double*)sh[r].d_lc, (double*)sh[r].d_gc, nullptr), shard_ctx_[r], "all-gather c*");
        }
        in_group = false;
        check(gpc_group_end(), shard_ctx_[0], "gpc_group_end");
        // device 0 un-permutes to patch order and hands the grids to the host flow (load_compressed reprojects them)
        check(gpc_unpermute_fstar_dev(shard_comm_[0], m, (const double*)sh[0].d_gf, (double*)sh[0].d_f), shard_ctx_[0], "un-permute f*");
        check(gpc_unpermute_fstar_dev(shard_comm_[0], 3 * m, (const double*)sh[0].d_gc, (double*)sh[0].d_c), shard_ctx_[0], "un-permute c*");
        dense_f_.assign((size_t)P * m, 0.0);
        dense_c_.assign((size_t)P * 3 * m, 0.0);
        check(gpc_dev_memcpy(shard_ctx_[0], dense_f_.data(), sh[0].d_f, 8 * (size_t)P * m, GPC_COPY_D2H), shard_ctx_[0], "download f*");
        check(gpc_dev_memcpy(shard_ctx_[0], dense_c_.data(), sh[0].d_c, 8 * (size_t)P * 3 * m, GPC_COPY_D2H), shard_ctx_[0], "download c*");
        std::vector<int32_t> st(S);
        for (int r = 0; r < world; ++r) {
            check(gpc_dev_memcpy(shard_ctx_[r], st.data(), sh[r].d_st, sizeof(int32_t) * (size_t)S, GPC_COPY_D2H), shard_ctx_[r], "download status");
            for (int q = 0; q < S; ++q) {
                const int p = slots[(size_t)r * S + q];
                if (p >= 0) status_[p] = st[q];
            }
        }
        for (int r = 0; r < world; ++r) check(gpc_ctx_synchronize(shard_ctx_[r]), shard_ctx_[r], "gpc_ctx_synchronize");
    } catch (...) {
        if (in_group) (void)gpc_group_end();
        for (int r = 0; r < world; ++r) (void)gpc_ctx_synchronize(shard_ctx_[r]);    // nothing may still read the buffers freed below
        free_all();
        throw;
    }
    free_all();
}

gp_compressor::~gp_compressor()
{
    for (gpc_sparse* q : shard_gps_) if (q) gpc_sparse_destroy(q);
    for (gpc_sparse* q : shard_rgb_) if (q) gpc_sparse_destroy(q);
    for (gpc_comm* c : shard_comm_) gpc_comm_destroy(c);
    for (gpc_ctx* c : shard_ctx_) gpc_ctx_destroy(c);
    release_device();
    if (gps_) gpc_sparse_destroy(gps_);
    if (rgb_gps_) gpc_sparse_destroy(rgb_gps_);
    if (ctx_) gpc_ctx_destroy(ctx_);
}

// src/gp_compressor.cpp:21-27
void gp_compressor::save_compressed(const std::string& /*name: ignored by the reference too*/)
{
    if (gpu_producer && devices_.empty()) project_cloud_device(); else project_cloud();
    train_processes();
}

// project_cloud() on the GPU: upload the cloud as pcl::PointXYZRGB records, fetch the batch
void gp_compressor::project_cloud_device()
{
    if (!ctx_) check(gpc_ctx_create(&ctx_, device_), nullptr, "gpc_ctx_create");
    std::vector<gpc_point_xyzrgb> rec(cloud_.size());
    for (size_t i = 0; i < cloud_.size(); ++i) {
        const point& p = cloud_[i];
        rec[i] = gpc_point_xyzrgb{p.x, p.y, p.z, 1.0f, p.b, p.g, p.r, 255, {0.0f, 0.0f, 0.0f}};
    }
    release_device();
    gpc_patches* pt = nullptr;
    check(gpc_project_cloud(ctx_, rec.data(), (int)rec.size(), res_, sz_, &pt), ctx_, "gpc_project_cloud");
    gpc_patches_view v;
    gpc_patches_view_dev(pt, &v);
    batch_ = patch_batch();
    const size_t P = (size_t)v.P, N = (size_t)v.n_total;
    batch_.off.assign(P + 1, 0);
    batch_.x0.resize(N); batch_.x1.resize(N); batch_.y.resize(N); batch_.rgb.resize(3 * N);
    batch_.rotations.resize(P); batch_.means.resize(P); batch_.rgb_means.resize(P);
    batch_.W.resize(P * (size_t)v.m);
    const int rc = gpc_patches_fetch(pt, batch_.off.data(), batch_.x0.data(), batch_.x1.data(), batch_.y.data(), batch_.rgb.data(),
                                     P ? batch_.rotations[0].data() : nullptr, P ? batch_.means[0].data() : nullptr,
                                     P ? batch_.rgb_means[0].data() : nullptr, batch_.W.data(), nullptr);
    if (rc != GPC_OK) gpc_patches_destroy(pt); else dev_patches_ = pt;     // the batch stays on the device for train_processes()
    check(rc, ctx_, "gpc_patches_fetch");
    projected_ = true;
}

// src/gp_compressor.cpp:29-64
void gp_compressor::compute_rotation(double R[9], const std::vector<double>& pts4, int k) const
{
    auto setcol = [&](int c, const double v[3]) { R[3 * c] = v[0]; R[3 * c + 1] = v[1]; R[3 * c + 2] = v[2]; };
    if (k < 4) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(R, I, sizeof(I));
        return;
    }
    double M[4][4] = {};
    for (int p = 0; p < k; ++p)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) M[a][b] += pts4[4 * p + a] * pts4[4 * p + b];
    double v[4];
    smallest_eigvec4(M, v);
    double normal[3] = {v[0], v[1], v[2]};
    normalize(normal);
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, z[3] = {0, 0, 1};
    double c1[3];
    const double ax = std::fabs(normal[0]), ay = std::fabs(normal[1]), az = std::fabs(normal[2]);
    if (ax > ay && ax > az) {            // pointing in x dir
        if (normal[0] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross(z, normal, c1);
    } else if (ay > ax && ay > az) {     // pointing in y dir
        if (normal[1] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross(x, normal, c1);
    } else {                             // pointing in z dir
        if (normal[2] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross(y, normal, c1);
    }
    normalize(c1);
    double c2[3];
    cross(normal, c1, c2);
    setcol(0, normal);
    setcol(1, c1);
    setcol(2, c2);
}

// src/gp_compressor.cpp:177-249 with project_points (:66-118) inlined per leaf
void gp_compressor::project_cloud()
{
    release_device();
    batch_ = patch_batch();
    batch_.off.push_back(0);
    const size_t npts = cloud_.size();
    if (npts == 0) { projected_ = true; return; }
    // voxel hash of side res, anchored at the cloud's minimum corner
    double mn[3] = {cloud_[0].x, cloud_[0].y, cloud_[0].z};
    for (const point& p : cloud_) {
        mn[0] = std::min<double>(mn[0], p.x);
        mn[1] = std::min<double>(mn[1], p.y);
        mn[2] = std::min<double>(mn[2], p.z);
    }
    auto key_of = [&](const point& p) {
        return VoxelKey{(int)std::floor((p.x - mn[0]) / res_), (int)std::floor((p.y - mn[1]) / res_),
                        (int)std::floor((p.z - mn[2]) / res_)};
    };
    std::unordered_map<VoxelKey, std::vector<int>, VoxelHash> grid;
    for (size_t i = 0; i < npts; ++i) grid[key_of(cloud_[i])].push_back((int)i);
    std::vector<VoxelKey> leaves;
    leaves.reserve(grid.size());
    for (auto& kv : grid) leaves.push_back(kv.first);
    std::sort(leaves.begin(), leaves.end());   // deterministic leaf order (the octree's depth-first order is PCL's)

    const double radius = std::sqrt(3.0f) / 2.0f * res_;   // :194, sphere encompassing the voxel
    std::vector<char> occupied(npts, 0);                   // occupied_indices (:200): exclusive point ownership
    std::vector<int> index_search;
    std::vector<double> pts4, cols;
    const int m = sz_ * sz_;
    for (const VoxelKey& key : leaves) {
        const double center[3] = {mn[0] + (key.x + 0.5) * res_, mn[1] + (key.y + 0.5) * res_, mn[2] + (key.z + 0.5) * res_};
        // radiusSearch(center, radius) (:220): radius < res, so the 27 neighbouring voxels cover the sphere
        index_search.clear();
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    auto it = grid.find(VoxelKey{key.x + dx, key.y + dy, key.z + dz});
                    if (it == grid.end()) continue;
                    for (int idx : it->second) {
                        const point& p = cloud_[idx];
                        const double ex = p.x - center[0], ey = p.y - center[1], ez = p.z - center[2];
                        if (ex * ex + ey * ey + ez * ez <= radius * radius) index_search.push_back(idx);
                    }
                }
        // hit order = patch point order: neighbour voxels in (dz, dy, dx) order, ascending index inside a voxel (PCL's is its
        // octree traversal order); the GPU producer and the oracle walk the same order, so all three agree bit for bit
        std::array<double, 9> R{};
        std::array<double, 3> mid{center[0], center[1], center[2]}, cmean{0, 0, 0};
        const int k = (int)index_search.size();
        pts4.assign(4 * (size_t)k, 1.0);
        cols.assign(3 * (size_t)k, 0.0);
        for (int q = 0; q < k; ++q) {
            const point& p = cloud_[index_search[q]];
            pts4[4 * q] = p.x; pts4[4 * q + 1] = p.y; pts4[4 * q + 2] = p.z;
            cols[3 * q] = p.r; cols[3 * q + 1] = p.g; cols[3 * q + 2] = p.b;
        }
        compute_rotation(R.data(), pts4, k);
        // project_points (:66-118)
        const size_t first = batch_.x0.size();
        std::vector<uint8_t> W(m, 0);
        double mnd = 0;
        std::vector<std::array<double, 3>> colours;
        for (int q = 0; q < k; ++q) {
            const int gi = index_search[q];
            if (occupied[gi]) continue;                                         // :81-83
            const double d[3] = {pts4[4 * q] - mid[0], pts4[4 * q + 1] - mid[1], pts4[4 * q + 2] - mid[2]};
            double pt[3];
            for (int a = 0; a < 3; ++a) pt[a] = R[3 * a] * d[0] + R[3 * a + 1] * d[1] + R[3 * a + 2] * d[2];   // R^T d
            if (pt[1] > res_ / 2.0f || pt[1] < -res_ / 2.0f || pt[2] > res_ / 2.0f || pt[2] < -res_ / 2.0f) continue;   // :85-87
            mnd += pt[0];
            occupied[gi] = 1;
            int gx = (int)((double)sz_ * (pt[1] / res_ + 0.5f)), gy = (int)((double)sz_ * (pt[2] / res_ + 0.5f));     // :90-92
            gx = std::min(std::max(gx, 0), sz_ - 1);
            gy = std::min(std::max(gy, 0), sz_ - 1);
            W[sz_ * gx + gy] = 1;
            batch_.y.push_back(pt[0]);
            batch_.x0.push_back(pt[1]);
            batch_.x1.push_back(pt[2]);
            colours.push_back({cols[3 * q], cols[3 * q + 1], cols[3 * q + 2]});
            for (int a = 0; a < 3; ++a) cmean[a] += cols[3 * q + a];
        }
        const size_t cnt = batch_.x0.size() - first;
        if (cnt > 0) {
            mnd /= (double)cnt;                                                 // :101-107
            for (int a = 0; a < 3; ++a) cmean[a] /= (double)cnt;
            for (size_t q = first; q < batch_.y.size(); ++q) batch_.y[q] -= mnd;
            for (int a = 0; a < 3; ++a) mid[a] += mnd * R[a];                   // center += mn*R.col(0)  (:116)
        }
        // colours are appended plane-wise after the loop (3 planes of N); keep them per patch for now
        for (auto& c : colours)
            for (int a = 0; a < 3; ++a) batch_.rgb.push_back(c[a] - cmean[a]);   // temporary AoS, re-packed below
        batch_.off.push_back((int32_t)batch_.x0.size());
        batch_.rotations.push_back(R);
        batch_.means.push_back(mid);
        batch_.rgb_means.push_back(cmean);
        batch_.W.insert(batch_.W.end(), W.begin(), W.end());
    }
    // AoS colours -> 3 planes of N (Eigen column-major n x 3, per the C-ABI layout)
    const size_t N = batch_.x0.size();
    std::vector<double> planes(3 * N);
    for (size_t q = 0; q < N; ++q)
        for (int a = 0; a < 3; ++a) planes[a * N + q] = batch_.rgb[3 * q + a];
    batch_.rgb.swap(planes);
    projected_ = true;
}

// sparse_gp::shuffle (src/sparse_gp.hpp:43-56): ind[i] <-> ind[rand() % i] for i = n-1 .. 1
void gp_compressor::shuffle(std::vector<int32_t>& perm, int n)
{
    const size_t base = perm.size();
    for (int i = 0; i < n; ++i) perm.push_back(i);
    for (int i = n - 1; i > 0; --i) {
        const int r = rng() % i;
        std::swap(perm[base + i], perm[base + r]);
    }
}

// src/gp_compressor.cpp:121-175, batched
void gp_compressor::train_processes()
{
    if (!projected_) project_cloud();
    const int P = batch_.patches();
    if (P == 0) { trained_ = true; return; }
    if (!ctx_) check(gpc_ctx_create(&ctx_, device_), nullptr, "gpc_ctx_create");
    const int m = sz_ * sz_;
    status_.assign(P, 0);
    gpc_patches_view dv{};
    const bool on_device = dev_patches_ != nullptr;
    if (on_device) gpc_patches_view_dev(dev_patches_, &dv);
    if (model_ == gp_model::dense && on_device) {
        // depth and colour grids predicted straight from the device batch; they stay in HBM for load_compressed()
        void* d_st = nullptr;
        if (d_dense_f_) { (void)gpc_dev_free(ctx_, d_dense_f_); d_dense_f_ = nullptr; }
        if (d_dense_c_) { (void)gpc_dev_free(ctx_, d_dense_c_); d_dense_c_ = nullptr; }
        check(gpc_dev_malloc(ctx_, sizeof(double) * (size_t)P * m, (void**)&d_dense_f_), ctx_, "gpc_dev_malloc");
        check(gpc_dev_malloc(ctx_, sizeof(double) * (size_t)P * 3 * m, (void**)&d_dense_c_), ctx_, "gpc_dev_malloc");
        check(gpc_dev_malloc(ctx_, sizeof(int32_t) * (size_t)P, &d_st), ctx_, "gpc_dev_malloc");
        int rc = gpc_dense_fit_predict_grid_dev(ctx_, &dense_params, P, dv.off, dv.n_max, dv.n_total, dv.x0, dv.x1, dv.y, 1, res_, sz_,
                                                d_dense_f_, nullptr, (int32_t*)d_st);
        if (rc == GPC_OK)
            rc = gpc_dense_fit_predict_grid_dev(ctx_, &dense_params, P, dv.off, dv.n_max, dv.n_total, dv.x0, dv.x1, dv.rgb, 3, res_, sz_,
                                                d_dense_c_, nullptr, (int32_t*)d_st);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, status_.data(), d_st, sizeof(int32_t) * (size_t)P, GPC_COPY_D2H);
        (void)gpc_dev_free(ctx_, d_st);
        check(rc, ctx_, "gpc_dense_fit_predict_grid_dev");
        mean_added_ = 0;
        max_added_ = 0;
        for (int i = 0; i < P; ++i) {
            const int n = batch_.off[i + 1] - batch_.off[i];
            mean_added_ += n;
            max_added_ = std::max(max_added_, n);
        }
        mean_added_ /= P;
        trained_ = true;
        return;
    }
    if (model_ == gp_model::dense) {
        if (!devices_.empty()) {
            train_dense_sharded();
        } else {
            dense_f_.assign((size_t)P * m, 0.0);
            dense_c_.assign((size_t)P * 3 * m, 0.0);
            check(gpc_dense_fit_predict_grid(ctx_, &dense_params, P, batch_.off.data(), batch_.x0.data(), batch_.x1.data(),
                                             batch_.y.data(), 1, res_, sz_, dense_f_.data(), nullptr, status_.data()),
                  ctx_, "gpc_dense_fit_predict_grid(depth)");
            check(gpc_dense_fit_predict_grid(ctx_, &dense_params, P, batch_.off.data(), batch_.x0.data(), batch_.x1.data(),
                                             batch_.rgb.data(), 3, res_, sz_, dense_c_.data(), nullptr, status_.data()),
                  ctx_, "gpc_dense_fit_predict_grid(rgb)");
        }
        mean_added_ = 0;
        max_added_ = 0;
        for (int i = 0; i < P; ++i) {
            const int n = batch_.off[i + 1] - batch_.off[i];
            mean_added_ += n;
            max_added_ = std::max(max_added_, n);
        }
        mean_added_ /= P;
        trained_ = true;
        return;
    }
    const bool sharded = !devices_.empty();
    if (!sharded) {
        if (!gps_) check(gpc_sparse_create(ctx_, &depth_params, P, 1, &gps_), ctx_, "gpc_sparse_create(depth)");
        if (!rgb_gps_) check(gpc_sparse_create(ctx_, &rgb_params, P, 3, &rgb_gps_), ctx_, "gpc_sparse_create(rgb)");
    }
    // the reference shuffles inside gps[i].add_measurements and again inside RGB_gps[i].add_measurements, patch by
    // patch (:162-163): draw the two orders in that interleaving
    std::vector<int32_t> perm_d, perm_c;
    for (int i = 0; i < P; ++i) {
        const int n = batch_.off[i + 1] - batch_.off[i];
        shuffle(perm_d, n);
        shuffle(perm_c, n);
    }
    if (sharded) {
        train_sparse_sharded(perm_d, perm_c);
    } else if (on_device) {
        // the points are already in HBM: only the two insertion orders go up
        const size_t N = (size_t)dv.n_total;
        void *d_pd = nullptr, *d_pc = nullptr, *d_st = nullptr;
        check(gpc_dev_malloc(ctx_, sizeof(int32_t) * N, &d_pd), ctx_, "gpc_dev_malloc");
        check(gpc_dev_malloc(ctx_, sizeof(int32_t) * N, &d_pc), ctx_, "gpc_dev_malloc");
        check(gpc_dev_malloc(ctx_, sizeof(int32_t) * (size_t)P, &d_st), ctx_, "gpc_dev_malloc");
        int rc = gpc_dev_memcpy(ctx_, d_pd, perm_d.data(), sizeof(int32_t) * N, GPC_COPY_H2D);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, d_pc, perm_c.data(), sizeof(int32_t) * N, GPC_COPY_H2D);
        if (rc == GPC_OK)
            rc = gpc_sparse_add_dev(gps_, dv.off, dv.n_max, dv.n_total, dv.x0, dv.x1, dv.y, (const int32_t*)d_pd, (int32_t*)d_st);
        if (rc == GPC_OK)
            rc = gpc_sparse_add_dev(rgb_gps_, dv.off, dv.n_max, dv.n_total, dv.x0, dv.x1, dv.rgb, (const int32_t*)d_pc, (int32_t*)d_st);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, status_.data(), d_st, sizeof(int32_t) * (size_t)P, GPC_COPY_D2H);
        (void)gpc_dev_free(ctx_, d_pd);
        (void)gpc_dev_free(ctx_, d_pc);
        (void)gpc_dev_free(ctx_, d_st);
        check(rc, ctx_, "gpc_sparse_add_dev");
    } else {
        check(gpc_sparse_add(gps_, batch_.off.data(), batch_.x0.data(), batch_.x1.data(), batch_.y.data(), perm_d.data(),
                             status_.data()), ctx_, "gpc_sparse_add(depth)");
        check(gpc_sparse_add(rgb_gps_, batch_.off.data(), batch_.x0.data(), batch_.x1.data(), batch_.rgb.data(), perm_c.data(),
                             status_.data()), ctx_, "gpc_sparse_add(rgb)");
    }
    std::vector<int32_t> bv(P, 0);
    if (sharded) {
        const int world = (int)devices_.size(), S = (P + world - 1) / world;
        std::vector<int32_t> sz_slot(S);
        for (int r = 0; r < world; ++r) {
            check(gpc_sparse_sizes(shard_gps_[r], sz_slot.data()), shard_ctx_[r], "gpc_sparse_sizes (shard)");
            for (int q = 0; q < S; ++q)
                if (shard_slots_[(size_t)r * S + q] >= 0) bv[shard_slots_[(size_t)r * S + q]] = sz_slot[q];
        }
    } else {
        check(gpc_sparse_sizes(gps_, bv.data()), ctx_, "gpc_sparse_sizes");
    }
    double mean = 0, added = 0;                  // "Mean added" / "Max added" (:164-168, 173-174)
    int maxm = 0;
    for (int i = 0; i < P; ++i) {
        if (batch_.off[i + 1] == batch_.off[i]) continue;
        mean = (added * mean + bv[i]) / (added + 1);
        maxm = std::max(maxm, (int)bv[i]);
        added += 1;
    }
    mean_added_ = mean;
    max_added_ = maxm;
    trained_ = true;
}

// src/gp_compressor.cpp:251-265 (x.cast<short>() as on x86-64, see oracle/gpc_oracle.c)
void gp_compressor::flatten_colors(uint8_t out[3], const double c[3])
{
    for (int i = 0; i < 3; ++i) {
        const double x = c[i];
        int v;
        if (std::isnan(x) || std::isinf(x)) {
            v = 255;
        } else {
            const int32_t w = (x >= 2147483648.0 || x < -2147483648.0) ? INT32_MIN : (int32_t)x;
            v = (int16_t)(uint16_t)(uint32_t)w;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        out[i] = (uint8_t)v;
    }
}

// src/gp_compressor.cpp:267-386
pointcloud gp_compressor::load_compressed()
{
    pointcloud out;
    if (!trained_) return out;
    const int P = batch_.patches();
    const int m = sz_ * sz_;
    if (P == 0) return out;
    std::vector<double> xs0(m), xs1(m);
    int pcount = 0;
    for (int y = 0; y < sz_; ++y)                     // :320-331, y outer / x inner
        for (int x = 0; x < sz_; ++x) {
            xs0[pcount] = res_ * (((double)x + 0.5f) / (double)sz_ - 0.5f);
            xs1[pcount] = res_ * (((double)y + 0.5f) / (double)sz_ - 0.5f);
            ++pcount;
        }
    if (dev_patches_) {
        // predict (sparse) and reproject on the device; the records are the only download
        gpc_patches_view dv{};
        gpc_patches_view_dev(dev_patches_, &dv);
        std::vector<int32_t> bvh(P, 1);
        if (model_ == gp_model::dense) for (int i = 0; i < P; ++i) bvh[i] = batch_.off[i + 1] - batch_.off[i];
        else check(gpc_sparse_sizes(gps_, bvh.data()), ctx_, "gpc_sparse_sizes");
        void *d_xs0 = nullptr, *d_xs1 = nullptr, *d_bv = nullptr, *d_f = nullptr, *d_c = nullptr, *d_rec = nullptr, *d_n = nullptr;
        auto free_all = [&]() { for (void* q : {d_xs0, d_xs1, d_bv, d_rec, d_n}) if (q) (void)gpc_dev_free(ctx_, q);
                                if (model_ != gp_model::dense) { if (d_f) (void)gpc_dev_free(ctx_, d_f); if (d_c) (void)gpc_dev_free(ctx_, d_c); } };
        int rc = gpc_dev_malloc(ctx_, 8 * (size_t)m, &d_xs0);
        if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, 8 * (size_t)m, &d_xs1);
        if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, 4 * (size_t)P, &d_bv);
        if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, sizeof(gpc_point_xyzrgb) * (size_t)P * m, &d_rec);
        if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, 4, &d_n);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, d_xs0, xs0.data(), 8 * (size_t)m, GPC_COPY_H2D);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, d_xs1, xs1.data(), 8 * (size_t)m, GPC_COPY_H2D);
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, d_bv, bvh.data(), 4 * (size_t)P, GPC_COPY_H2D);
        if (model_ == gp_model::dense) {
            d_f = d_dense_f_;
            d_c = d_dense_c_;
        } else {
            if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, 8 * (size_t)P * m, &d_f);
            if (rc == GPC_OK) rc = gpc_dev_malloc(ctx_, 8 * (size_t)P * 3 * m, &d_c);
            if (rc == GPC_OK) rc = gpc_sparse_predict_dev(gps_, m, (const double*)d_xs0, (const double*)d_xs1, (double*)d_f, nullptr, 0, nullptr);
            if (rc == GPC_OK) rc = gpc_sparse_predict_dev(rgb_gps_, m, (const double*)d_xs0, (const double*)d_xs1, (double*)d_c, nullptr, 0, nullptr);
        }
        if (rc == GPC_OK)
            rc = gpc_reproject_dev(ctx_, P, m, (const int32_t*)d_bv, (const double*)d_xs0, (const double*)d_xs1, (const double*)d_f,
                                   (const double*)d_c, dv.rotations, dv.means, dv.rgb_means, (gpc_point_xyzrgb*)d_rec, (int32_t*)d_n);
        int32_t npts = 0;
        if (rc == GPC_OK) rc = gpc_dev_memcpy(ctx_, &npts, d_n, 4, GPC_COPY_D2H);
        std::vector<gpc_point_xyzrgb> recs((size_t)(rc == GPC_OK ? npts : 0));
        if (rc == GPC_OK && npts > 0) rc = gpc_dev_memcpy(ctx_, recs.data(), d_rec, sizeof(gpc_point_xyzrgb) * (size_t)npts, GPC_COPY_D2H);
        free_all();
        check(rc, ctx_, "load_compressed (device)");
        out.resize(recs.size());
        for (size_t q = 0; q < recs.size(); ++q) out[q] = point{recs[q].x, recs[q].y, recs[q].z, recs[q].r, recs[q].g, recs[q].b};
        return out;
    }
    std::vector<double> f_star, c_star;
    std::vector<int32_t> bv(P, 1);
    if (model_ == gp_model::dense) {
        f_star = dense_f_;
        c_star = dense_c_;
        for (int i = 0; i < P; ++i) bv[i] = batch_.off[i + 1] - batch_.off[i];
    } else if (!devices_.empty()) {
        predict_sparse_sharded(xs0, xs1, f_star, c_star, bv);
    } else {
        f_star.assign((size_t)P * m, 0.0);
        c_star.assign((size_t)P * 3 * m, 0.0);
        check(gpc_sparse_sizes(gps_, bv.data()), ctx_, "gpc_sparse_sizes");
        // the caller discards V_star (:333-334), so sigma is not requested
        check(gpc_sparse_predict(gps_, m, xs0.data(), xs1.data(), f_star.data(), nullptr, 0, nullptr), ctx_, "gpc_sparse_predict(depth)");
        check(gpc_sparse_predict(rgb_gps_, m, xs0.data(), xs1.data(), c_star.data(), nullptr, 0, nullptr), ctx_, "gpc_sparse_predict(rgb)");
    }
    // reprojection + colour clamp (:335-373) fused on the GPU: the predicted grids become pcl::PointXYZRGB records
    std::vector<double> R((size_t)P * 9), mu((size_t)P * 3), cm((size_t)P * 3);
    for (int i = 0; i < P; ++i) {
        std::memcpy(&R[(size_t)i * 9], batch_.rotations[i].data(), 9 * sizeof(double));
        std::memcpy(&mu[(size_t)i * 3], batch_.means[i].data(), 3 * sizeof(double));
        std::memcpy(&cm[(size_t)i * 3], batch_.rgb_means[i].data(), 3 * sizeof(double));
    }
    std::vector<gpc_point_xyzrgb> recs((size_t)P * m);
    int32_t npts = 0;
    check(gpc_reproject(ctx_, P, m, bv.data(), xs0.data(), xs1.data(), f_star.data(), c_star.data(), R.data(), mu.data(), cm.data(),
                        recs.data(), &npts), ctx_, "gpc_reproject");
    out.resize((size_t)npts);
    for (int q = 0; q < npts; ++q) out[q] = point{recs[q].x, recs[q].y, recs[q].z, recs[q].r, recs[q].g, recs[q].b};
    return out;
}

// ---- model file (row f3) ---------------------------------------------------------------------------------------------
// little-endian; header: "GPCM", u32 version = 1, f64 res, i32 sz, i32 P, gpc_params depth, gpc_params rgb (raw structs, with their
// size in front); per patch: f64 R[9], mean[3], rgb_mean[3]; i32 b_depth, b_rgb; f64 BV_depth[b][2], alpha_depth[b],
// BV_rgb[b'][2], alpha_rgb[3][b'].
namespace {
struct file_closer { void operator()(FILE* f) const { if (f) std::fclose(f); } };
template <class T> void wr(FILE* f, const T* p, size_t n) { if (n && std::fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("model file: write failed"); }
template <class T> void rd(FILE* f, T* p, size_t n) { if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("model file: truncated"); }
}  // namespace

gp_compressor::gp_compressor(double res, int sz, int device)
    : rng([] { return std::rand(); }), res_(res), sz_(sz), model_(gp_model::sparse), device_(device)
{
    gpc_default_params_sparse(&depth_params, 1);
    gpc_default_params_sparse(&rgb_params, 3);
    gpc_default_params_dense(&dense_params);
}

size_t gp_compressor::save_model(const std::string& path)
{
    if (model_ != gp_model::sparse) throw std::runtime_error("save_model: the sparse model only");
    if (!devices_.empty()) throw std::runtime_error("save_model: the single-device flow only (the sharded states live on their devices)");
    if (!trained_) { project_cloud(); train_processes(); }
    const int P = batch_.patches();
    std::unique_ptr<FILE, file_closer> f(std::fopen(path.c_str(), "wb"));
    if (!f) throw std::runtime_error("save_model: cannot open " + path);
    const uint32_t version = 1, psz = (uint32_t)sizeof(gpc_params);
    wr(f.get(), "GPCM", 4);
    wr(f.get(), &version, 1);
    wr(f.get(), &res_, 1);
    const int32_t hdr[2] = {sz_, P};
    wr(f.get(), hdr, 2);
    wr(f.get(), &psz, 1);
    wr(f.get(), &depth_params, 1);
    wr(f.get(), &rgb_params, 1);
    if (P > 0) {
        const int ldd = gpc_sparse_ld(gps_), ldc = gpc_sparse_ld(rgb_gps_);
        std::vector<int32_t> bd(P), bc(P);
        check(gpc_sparse_sizes(gps_, bd.data()), ctx_, "gpc_sparse_sizes");
        check(gpc_sparse_sizes(rgb_gps_, bc.data()), ctx_, "gpc_sparse_sizes");
        std::vector<double> ad((size_t)P * ldd), bvd((size_t)P * ldd * 2), ac((size_t)P * 3 * ldc), bvc((size_t)P * ldc * 2);
        check(gpc_sparse_get_state(gps_, ad.data(), nullptr, nullptr, bvd.data()), ctx_, "gpc_sparse_get_state");
        check(gpc_sparse_get_state(rgb_gps_, ac.data(), nullptr, nullptr, bvc.data()), ctx_, "gpc_sparse_get_state");
        for (int i = 0; i < P; ++i) {
            wr(f.get(), batch_.rotations[i].data(), 9);
            wr(f.get(), batch_.means[i].data(), 3);
            wr(f.get(), batch_.rgb_means[i].data(), 3);
            const int32_t b2[2] = {bd[i], bc[i]};
            wr(f.get(), b2, 2);
            wr(f.get(), &bvd[(size_t)i * ldd * 2], (size_t)2 * bd[i]);
            wr(f.get(), &ad[(size_t)i * ldd], (size_t)bd[i]);
            wr(f.get(), &bvc[(size_t)i * ldc * 2], (size_t)2 * bc[i]);
            for (int c = 0; c < 3; ++c) wr(f.get(), &ac[((size_t)i * 3 + c) * ldc], (size_t)bc[i]);
        }
    }
    const long pos = std::ftell(f.get());
    return pos < 0 ? 0 : (size_t)pos;
}

gp_compressor* gp_compressor::load_model(const std::string& path, int device)
{
    std::unique_ptr<FILE, file_closer> f(std::fopen(path.c_str(), "rb"));
    if (!f) throw std::runtime_error("load_model: cannot open " + path);
    char magic[4];
    uint32_t version = 0, psz = 0;
    double res = 0;
    int32_t hdr[2] = {0, 0};
    rd(f.get(), magic, 4);
    rd(f.get(), &version, 1);
    if (std::memcmp(magic, "GPCM", 4) != 0 || version != 1) throw std::runtime_error("load_model: not a GPCM v1 file");
    rd(f.get(), &res, 1);
    rd(f.get(), hdr, 2);
    rd(f.get(), &psz, 1);
    if (psz != sizeof(gpc_params) || hdr[0] <= 0 || hdr[1] < 0) throw std::runtime_error("load_model: bad header");
    std::unique_ptr<gp_compressor> g(new gp_compressor(res, hdr[0], device));
    rd(f.get(), &g->depth_params, 1);
    rd(f.get(), &g->rgb_params, 1);
    const int P = hdr[1];
    g->batch_.off.assign((size_t)P + 1, 0);
    g->batch_.rotations.resize(P);
    g->batch_.means.resize(P);
    g->batch_.rgb_means.resize(P);
    g->projected_ = true;
    if (P > 0) {
        check(gpc_ctx_create(&g->ctx_, device), nullptr, "gpc_ctx_create");
        check(gpc_sparse_create(g->ctx_, &g->depth_params, P, 1, &g->gps_), g->ctx_, "gpc_sparse_create(depth)");
        check(gpc_sparse_create(g->ctx_, &g->rgb_params, P, 3, &g->rgb_gps_), g->ctx_, "gpc_sparse_create(rgb)");
        const int ldd = gpc_sparse_ld(g->gps_), ldc = gpc_sparse_ld(g->rgb_gps_);
        std::vector<int32_t> bd(P), bc(P);
        std::vector<double> ad((size_t)P * ldd, 0.0), bvd((size_t)P * ldd * 2, 0.0), ac((size_t)P * 3 * ldc, 0.0), bvc((size_t)P * ldc * 2, 0.0);
        for (int i = 0; i < P; ++i) {
            rd(f.get(), g->batch_.rotations[i].data(), 9);
            rd(f.get(), g->batch_.means[i].data(), 3);
            rd(f.get(), g->batch_.rgb_means[i].data(), 3);
            int32_t b2[2];
            rd(f.get(), b2, 2);
            if (b2[0] < 0 || b2[0] > ldd || b2[1] < 0 || b2[1] > ldc) throw std::runtime_error("load_model: basis-vector count out of range");
            bd[i] = b2[0];
            bc[i] = b2[1];
            rd(f.get(), &bvd[(size_t)i * ldd * 2], (size_t)2 * bd[i]);
            rd(f.get(), &ad[(size_t)i * ldd], (size_t)bd[i]);
            rd(f.get(), &bvc[(size_t)i * ldc * 2], (size_t)2 * bc[i]);
            for (int c = 0; c < 3; ++c) rd(f.get(), &ac[((size_t)i * 3 + c) * ldc], (size_t)bc[i]);
        }
        check(gpc_sparse_set_state(g->gps_, bd.data(), ad.data(), nullptr, nullptr, bvd.data()), g->ctx_, "gpc_sparse_set_state(depth)");
        check(gpc_sparse_set_state(g->rgb_gps_, bc.data(), ac.data(), nullptr, nullptr, bvc.data()), g->ctx_, "gpc_sparse_set_state(rgb)");
    }
    g->trained_ = true;
    return g.release();
}

}  // namespace gpc

// ---- C doors for the Python test-suite -----------------------------------------------------------------------------
extern "C" {

// host-only: cut and project the patches of a cloud (no GPU needed).  Returns an opaque handle.
void* gpc_host_create(const float* xyz, const uint8_t* rgb, int n, double res, int sz, int model, int device)
{
    gpc::pointcloud c((size_t)n);
    for (int i = 0; i < n; ++i) c[i] = gpc::point{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    return new gpc::gp_compressor(c, res, sz, model == 1 ? gpc::gp_model::dense : gpc::gp_model::sparse, device);
}
void gpc_host_destroy(void* h) { delete static_cast<gpc::gp_compressor*>(h); }
void gpc_host_seed(void* h, unsigned seed)
{
    // deterministic stand-in for libc rand(): a 31-bit LCG owned by the object
    auto* g = static_cast<gpc::gp_compressor*>(h);
    auto state = std::make_shared<uint64_t>(seed);
    g->rng = [state]() -> int {
        *state = (*state * 6364136223846793005ULL + 1442695040888963407ULL);
        return (int)((*state >> 33) & 0x7fffffff);
    };
}
void gpc_host_set_sparse_kernel(void* h, double sigmaf_sq, double l_sq, double s20_depth, double s20_rgb, int capacity)
{
    auto* g = static_cast<gpc::gp_compressor*>(h);
    g->depth_params.sigmaf_sq = g->rgb_params.sigmaf_sq = sigmaf_sq;
    g->depth_params.l_sq = g->rgb_params.l_sq = l_sq;
    g->depth_params.noise = s20_depth;
    g->rgb_params.noise = s20_rgb;
    g->depth_params.capacity = g->rgb_params.capacity = capacity;
}
int gpc_host_project(void* h)
{
    try { static_cast<gpc::gp_compressor*>(h)->project_cloud(); return 0; } catch (...) { return -1; }
}
int gpc_host_project_device(void* h, char* err, int errlen)
{
    try { static_cast<gpc::gp_compressor*>(h)->project_cloud_device(); return 0; }
    catch (const std::exception& e) { if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.what()); return -1; }
}
void gpc_host_set_devices(void* h, const int* devices, int n)
{
    static_cast<gpc::gp_compressor*>(h)->set_devices(std::vector<int>(devices, devices + (n > 0 ? n : 0)));
}
void gpc_host_set_gpu_producer(void* h, int on) { static_cast<gpc::gp_compressor*>(h)->gpu_producer = on != 0; }
int gpc_host_patch_count(void* h) { return static_cast<gpc::gp_compressor*>(h)->patches().patches(); }
int gpc_host_point_count(void* h) { return (int)static_cast<gpc::gp_compressor*>(h)->patches().x0.size(); }
// copies the batch: off[P+1], x0/x1/y[N], rgb[3N], R[9P], mean[3P], rgb_mean[3P]
void gpc_host_get_batch(void* h, int32_t* off, double* x0, double* x1, double* y, double* rgb, double* R, double* mean,
                        double* rgb_mean)
{
    const gpc::patch_batch& b = static_cast<gpc::gp_compressor*>(h)->patches();
    std::memcpy(off, b.off.data(), sizeof(int32_t) * b.off.size());
    const size_t N = b.x0.size(), P = (size_t)b.patches();
    std::memcpy(x0, b.x0.data(), 8 * N);
    std::memcpy(x1, b.x1.data(), 8 * N);
    std::memcpy(y, b.y.data(), 8 * N);
    std::memcpy(rgb, b.rgb.data(), 8 * 3 * N);
    for (size_t i = 0; i < P; ++i) {
        std::memcpy(R + 9 * i, b.rotations[i].data(), 72);
        std::memcpy(mean + 3 * i, b.means[i].data(), 24);
        std::memcpy(rgb_mean + 3 * i, b.rgb_means[i].data(), 24);
    }
}
// GPU: save_compressed + load_compressed.  out_xyz / out_rgb hold up to P*sz*sz points; returns the count or < 0.
void gpc_host_get_mask(void* h, uint8_t* W)
{
    const auto& b = static_cast<gpc::gp_compressor*>(h)->patches();
    if (!b.W.empty()) std::memcpy(W, b.W.data(), b.W.size());
}

int gpc_host_roundtrip(void* h, float* out_xyz, uint8_t* out_rgb, int capacity_pts, double* mean_added, int* max_added,
                       char* err, int errlen)
{
    auto* g = static_cast<gpc::gp_compressor*>(h);
    try {
        g->save_compressed("unused");
        gpc::pointcloud c = g->load_compressed();
        if ((int)c.size() > capacity_pts) return -2;
        for (size_t i = 0; i < c.size(); ++i) {
            out_xyz[3 * i] = c[i].x; out_xyz[3 * i + 1] = c[i].y; out_xyz[3 * i + 2] = c[i].z;
            out_rgb[3 * i] = c[i].r; out_rgb[3 * i + 1] = c[i].g; out_rgb[3 * i + 2] = c[i].b;
        }
        if (mean_added) *mean_added = g->mean_added();
        if (max_added) *max_added = g->max_added();
        return (int)c.size();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

// model file: write the trained sparse model / reconstruct a cloud from a file alone.  Return bytes / point count, < 0 on error.
long long gpc_host_save_model(void* h, const char* path, char* err, int errlen)
{
    try {
        return (long long)static_cast<gpc::gp_compressor*>(h)->save_model(path);
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}
int gpc_host_decompress_file(const char* path, int device, float* out_xyz, uint8_t* out_rgb, int capacity_pts, char* err, int errlen)
{
    try {
        std::unique_ptr<gpc::gp_compressor> g(gpc::gp_compressor::load_model(path, device));
        gpc::pointcloud c = g->load_compressed();
        if ((int)c.size() > capacity_pts) return -2;
        for (size_t i = 0; i < c.size(); ++i) {
            out_xyz[3 * i] = c[i].x; out_xyz[3 * i + 1] = c[i].y; out_xyz[3 * i + 2] = c[i].z;
            out_rgb[3 * i] = c[i].r; out_rgb[3 * i + 1] = c[i].g; out_rgb[3 * i + 2] = c[i].b;
        }
        return (int)c.size();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

}  // extern "C"
