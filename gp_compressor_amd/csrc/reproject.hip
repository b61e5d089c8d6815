// reproject.hip -- the step after the hot path: the tail of the patch loop of
// gp_compressor::load_compressed (/root/reference/src/gp_compressor.cpp:335-373) fused into one kernel that emits the
// reconstructed cloud as 32-byte pcl::PointXYZRGB records straight from the predicted grids:
//     pt  = R_i * (f*, x*_0, x*_1) + mean_i                  (:335-340), stored as float
//     rgb = flatten_colors(C*_row + RGB_mean_i)              (:367-372, :251-265)
// Patches whose GP holds no basis vector are skipped and the output is compacted in patch order, like the reference's
// running `counter` (:299-301).  Arithmetic is written with explicit round-to-nearest mul/add (no FMA contraction) in
// the reference's association, so xyz and rgb are bit-identical to the CPU oracle.
#include "gpc_device.h"
#include "gpc_internal.h"

#define RP_THREADS 256
#define RP_SCAN_THREADS 1024

// base[i] = m * #{ j < i : bv[j] != 0 } ; total[0] = m * #{ j : bv[j] != 0 }   (bv == nullptr: every patch is trained)
__global__ __launch_bounds__(RP_SCAN_THREADS) void reproject_scan_kernel(int P, int m, const int32_t* bv, int32_t* base, int32_t* total)
{
    __shared__ int part[RP_SCAN_THREADS];
    const int tid = threadIdx.x;
    const int per = (P + RP_SCAN_THREADS - 1) / RP_SCAN_THREADS;
    const int lo = min(P, tid * per), hi = min(P, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += (!bv || bv[i] != 0) ? 1 : 0;
    part[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < RP_SCAN_THREADS; o <<= 1) {      // inclusive Hillis-Steele scan
        const int v = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - cnt;                           // exclusive prefix of this thread's segment
    for (int i = lo; i < hi; ++i) {
        const bool on = !bv || bv[i] != 0;
        base[i] = on ? run * m : -1;
        run += on ? 1 : 0;
    }
    if (tid == RP_SCAN_THREADS - 1) total[0] = part[tid] * m;
}

// x.cast<short>() as the x86-64 reference binary evaluates it (cvttsd2si, then truncation to 16 bits), then the clamp
__device__ static inline uint8_t rp_flatten(double x)
{
    if (x != x || __builtin_isinf(x)) return 255;
    const int w = (x >= 2147483648.0 || x < -2147483648.0) ? (int)0x80000000 : (int)x;     // (int)x truncates toward zero
    const int v = (int)(short)(unsigned short)(unsigned)w;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct ReprojParams {
    int P, m;
    const double *xs0, *xs1, *f_star, *c_star, *R, *means, *rgb_means;
    const int32_t* base;
    gpc_point_xyzrgb* cloud;
};

__global__ __launch_bounds__(RP_THREADS) void reproject_kernel(ReprojParams A)
{
    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int b0 = A.base[patch];
        if (b0 < 0) continue;
        const double* R = A.R + (size_t)patch * 9;
        const double* mean = A.means + (size_t)patch * 3;
        for (int q = threadIdx.x; q < A.m; q += RP_THREADS) {
            const double f = A.f_star[(size_t)patch * A.m + q], a = A.xs0[q], b = A.xs1[q];
            gpc_point_xyzrgb p;
            float xyz[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double v = __dadd_rn(__dadd_rn(__dmul_rn(R[i], f), __dmul_rn(R[i + 3], a)), __dmul_rn(R[i + 6], b));
                xyz[i] = (float)__dadd_rn(v, mean[i]);
            }
            p.x = xyz[0]; p.y = xyz[1]; p.z = xyz[2]; p.w = 1.0f;
            uint8_t rgb[3] = {0, 0, 0};
            if (A.c_star) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    rgb[c] = rp_flatten(__dadd_rn(A.c_star[((size_t)patch * 3 + c) * A.m + q], A.rgb_means[(size_t)patch * 3 + c]));
            }
            p.r = rgb[0]; p.g = rgb[1]; p.b = rgb[2]; p.a = 255;
            p.pad[0] = p.pad[1] = p.pad[2] = 0.0f;
            A.cloud[(size_t)b0 + q] = p;
        }
    }
}

extern "C" {

int gpc_reproject_dev(gpc_ctx* ctx, int P, int m, const int32_t* bv_count, const double* xs0, const double* xs1,
                      const double* f_star, const double* c_star, const double* rotations, const double* means,
                      const double* rgb_means, gpc_point_xyzrgb* cloud, int32_t* n_points)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (P < 0 || m < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (!n_points) return gpc_fail(ctx, GPC_EINVAL, "n_points is NULL");
    if (P > 0 && m > 0 && (!xs0 || !xs1 || !f_star || !rotations || !means || !cloud))
        return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1/f_star/rotations/means/cloud is NULL");
    if (c_star && !rgb_means) return gpc_fail(ctx, GPC_EINVAL, "c_star needs rgb_means");
    if ((long long)P * (long long)m > 0x7fffffffLL) return gpc_fail(ctx, GPC_ERANGE, "P*m exceeds 2^31-1 points");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    if (P == 0 || m == 0) {
        GPC_HIP(ctx, hipMemsetAsync(n_points, 0, sizeof(int32_t), ctx->stream));
        return GPC_OK;
    }
    int rc = gpc_ws_reserve(ctx, sizeof(int32_t) * (size_t)P);
    if (rc != GPC_OK) return rc;
    int32_t* base = static_cast<int32_t*>(ctx->ws);
    hipLaunchKernelGGL(reproject_scan_kernel, dim3(1), dim3(RP_SCAN_THREADS), 0, ctx->stream, P, m, bv_count, base, n_points);
    GPC_HIP(ctx, hipGetLastError());
    ReprojParams A;
    A.P = P; A.m = m; A.xs0 = xs0; A.xs1 = xs1; A.f_star = f_star; A.c_star = c_star; A.R = rotations; A.means = means;
    A.rgb_means = rgb_means; A.base = base; A.cloud = cloud;
    const int grid = P < ctx->num_cus * 8 ? P : ctx->num_cus * 8;
    hipLaunchKernelGGL(reproject_kernel, dim3(grid), dim3(RP_THREADS), 0, ctx->stream, A);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

int gpc_reproject(gpc_ctx* ctx, int P, int m, const int32_t* bv_count, const double* xs0, const double* xs1, const double* f_star,
                  const double* c_star, const double* rotations, const double* means, const double* rgb_means,
                  gpc_point_xyzrgb* cloud, int32_t* n_points)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (P < 0 || m < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (!n_points) return gpc_fail(ctx, GPC_EINVAL, "n_points is NULL");
    *n_points = 0;
    if (P == 0 || m == 0) return GPC_OK;
    if (!xs0 || !xs1 || !f_star || !rotations || !means || !cloud) return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1/f_star/rotations/means/cloud is NULL");
    if (c_star && !rgb_means) return gpc_fail(ctx, GPC_EINVAL, "c_star needs rgb_means");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = gpc_stream_of(ctx);
    const size_t Pm = (size_t)P * m;
    void *d_bv = nullptr, *d_xs0 = nullptr, *d_xs1 = nullptr, *d_f = nullptr, *d_c = nullptr, *d_R = nullptr, *d_mu = nullptr,
         *d_cm = nullptr, *d_cloud = nullptr, *d_n = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_bv, d_xs0, d_xs1, d_f, d_c, d_R, d_mu, d_cm, d_cloud, d_n})
            if (p) (void)hipFree(p);
    };
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, s);
        return e;
    };
    hipError_t e = hipSuccess;
    if (bv_count) e = up(&d_bv, bv_count, 4 * (size_t)P);
    if (e == hipSuccess) e = up(&d_xs0, xs0, 8 * (size_t)m);
    if (e == hipSuccess) e = up(&d_xs1, xs1, 8 * (size_t)m);
    if (e == hipSuccess) e = up(&d_f, f_star, 8 * Pm);
    if (e == hipSuccess && c_star) e = up(&d_c, c_star, 8 * Pm * 3);
    if (e == hipSuccess) e = up(&d_R, rotations, 8 * (size_t)P * 9);
    if (e == hipSuccess) e = up(&d_mu, means, 8 * (size_t)P * 3);
    if (e == hipSuccess && c_star) e = up(&d_cm, rgb_means, 8 * (size_t)P * 3);
    if (e == hipSuccess) e = hipMalloc(&d_cloud, sizeof(gpc_point_xyzrgb) * Pm);
    if (e == hipSuccess) e = hipMalloc(&d_n, sizeof(int32_t));
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_reproject: %s", hipGetErrorString(e));
    }
    int rc = gpc_reproject_dev(ctx, P, m, (const int32_t*)d_bv, (const double*)d_xs0, (const double*)d_xs1, (const double*)d_f,
                               (const double*)d_c, (const double*)d_R, (const double*)d_mu, (const double*)d_cm,
                               (gpc_point_xyzrgb*)d_cloud, (int32_t*)d_n);
    if (rc == GPC_OK) {
        e = hipMemcpyAsync(n_points, d_n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess && *n_points > 0)
            e = hipMemcpy(cloud, d_cloud, sizeof(gpc_point_xyzrgb) * (size_t)*n_points, hipMemcpyDeviceToHost);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return gpc_fail(ctx, GPC_EHIP, "gpc_reproject: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

}  // extern "C"
