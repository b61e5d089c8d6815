// producer.hip -- the step before the hot path: gp_compressor::project_cloud with
// compute_rotation and project_points (/root/reference/src/gp_compressor.cpp:177-249, 29-64, 66-118) on the GPU.
// A pcl::PointXYZRGB cloud goes in; the ragged patch batch the GP kernels consume (off, X, y, C, R_i, mean_i, RGB mean,
// W) comes out resident in HBM, so that cloud -> patches -> GP -> cloud needs no host pass over the points.
//
// The reference walks the leaves of a PCL octree one after the other and lets each leaf claim the points of its
// search sphere that nobody claimed before (occupied_indices): a serial dependence in its letter, not in its substance --
//   * the frame R_i of a leaf depends on every point of its sphere, claimed or not (:220-236), so frames are independent;
//   * a point is claimed by the FIRST leaf in leaf order whose sphere holds it and whose +-res/2 window accepts it
//     (:81-89); candidates are the <= 27 leaves around the point's voxel, so ownership is a per-point minimum.
// Pipeline (all HBM-bound integer / gather work; nothing here is GEMM-shaped):
//   1 pc_bounds_kernel    min / max corner, finiteness: one pass over the cloud, one set of atomics per workgroup
//   2 pc_keys_kernel      voxel key (z, y, x packed) per point, then rocPRIM's stable radix sort of (key, index) over
//                         exactly the key's bits, pc_heads/pc_leaves: leaf table (sorted unique keys + segment starts)
//   3 pc_gather_kernel    points re-laid in sorted order as 16-byte records (x, y, z, rgb): every later access to a
//                         voxel's points is one contiguous, coalesced segment
//   4 pc_moment_kernel    one wave per leaf: 27 neighbour segments (binary search in the leaf table), sphere test by
//                         ballot; the hits' exact products go to LDS in hit order and 16 lanes (one matrix entry each)
//                         add them up in the oracle's order -- the only serial chain is one f64 add per hit
//     pc_frame_kernel     one thread per leaf: cyclic Jacobi eigen-solve of the 4x4 moment matrix, frame (:37-63)
//   5 pc_claim_kernel     one wave per leaf: the 27 candidate frames sit in LDS, the leaf's own points test them in
//                         leaf order and stop at the first that accepts; patch-frame coordinates; counts
//   6 pc_emit_kernel      one wave per leaf: ordered compaction (ballot + prefix popcount) of the points it owns, depth
//                         mean as the oracle's sequential sum, colour means, mean removal, centre shift, mask W
// Scratch lives in the context's grow-only workspace and the result in one allocation: a call costs three small
// device->host reads (bounds, leaf count, totals) and no allocation in steady state beyond the result's.
// Bit-exactness: floating-point contraction is off for this file and every expression is written in the association of
// oracle/gpc_oracle_producer.c; products of two floats are exact in double, so the moment sums only fix the ORDER.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gpc_internal.h"

#pragma clang fp contract(off)

#define PC_THREADS 256
#define PC_WAVES (PC_THREADS / 64)

struct PcGrid {
    double mn[3];       // minimum corner (the voxel grid's anchor)
    double res, radius, half;
    int kmax[3];        // largest voxel coordinate per axis
    int bx, by, bz;     // key = kz << (bx + by) | ky << bx | kx
    int sz;
};

struct PcPoint {        // sorted-order record
    float x, y, z;
    uint32_t rgb;       // r | g << 8 | b << 16
};

// ---- 1: bounds -------------------------------------------------------------------------------------------------------
__device__ static inline uint32_t pc_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float pc_unordered(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// out[0..2] = ordered min, out[3..5] = ordered max, out[6] = 1 if a coordinate is not finite
__global__ __launch_bounds__(PC_THREADS) void pc_bounds_kernel(const gpc_point_xyzrgb* cloud, int n, uint32_t* out)
{
    __shared__ uint32_t red[PC_WAVES][8];
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0, 0, 0};
    int bad = 0;
    for (int i = blockIdx.x * PC_THREADS + threadIdx.x; i < n; i += gridDim.x * PC_THREADS) {
        const float4 p = *reinterpret_cast<const float4*>(&cloud[i]);
        const float c[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bad |= !(fabsf(c[a]) <= 3.4028234e38f);
            const uint32_t o = pc_ordered(c[a]);
            lo[a] = min(lo[a], o);
            hi[a] = max(hi[a], o);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], o));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], o));
        }
    }
    bad = __any(bad);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[w][a] = lo[a]; red[w][3 + a] = hi[a]; }
        red[w][6] = (uint32_t)bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        uint32_t v = red[0][a];
        for (int q = 1; q < PC_WAVES; ++q) v = a < 3 ? min(v, red[q][a]) : max(v, red[q][a]);   // [6]: 0 / 1, max == or
        if (a < 3) atomicMin(&out[a], v); else if (a < 6) atomicMax(&out[a], v); else if (v) atomicOr(&out[6], 1u);
    }
}

// ---- 2: keys, leaf table -----------------------------------------------------------------------------------------------
__device__ static inline void pc_voxel(const PcGrid& g, float x, float y, float z, int k[3])
{
    k[0] = (int)floor(((double)x - g.mn[0]) / g.res);
    k[1] = (int)floor(((double)y - g.mn[1]) / g.res);
    k[2] = (int)floor(((double)z - g.mn[2]) / g.res);
}
__device__ static inline uint64_t pc_pack(const PcGrid& g, int kx, int ky, int kz)
{
    return ((uint64_t)kz << (g.bx + g.by)) | ((uint64_t)ky << g.bx) | (uint64_t)kx;
}
__device__ static inline void pc_unpack(const PcGrid& g, uint64_t key, int k[3])
{
    k[0] = (int)(key & ((1ull << g.bx) - 1));
    k[1] = (int)((key >> g.bx) & ((1ull << g.by) - 1));
    k[2] = (int)(key >> (g.bx + g.by));
}
__device__ static inline void pc_center(const PcGrid& g, const int k[3], double c[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = g.mn[a] + ((double)k[a] + 0.5) * g.res;
}

__global__ __launch_bounds__(PC_THREADS) void pc_keys_kernel(PcGrid g, const gpc_point_xyzrgb* cloud, int n, uint64_t* keys, int32_t* vals)
{
    const int i = blockIdx.x * PC_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 p = *reinterpret_cast<const float4*>(&cloud[i]);
    int k[3];
    pc_voxel(g, p.x, p.y, p.z, k);
    keys[i] = pc_pack(g, k[0], k[1], k[2]);
    vals[i] = i;
}

__global__ __launch_bounds__(PC_THREADS) void pc_heads_kernel(const uint64_t* keys, int n, int32_t* head)
{
    const int s = blockIdx.x * PC_THREADS + threadIdx.x;
    if (s < n) head[s] = (s == 0 || keys[s] != keys[s - 1]) ? 1 : 0;
}

// leaf_of[s] holds the inclusive scan of head[] on entry (leaf id + 1) and the leaf id on exit
__global__ __launch_bounds__(PC_THREADS) void pc_leaves_kernel(const uint64_t* keys, int n, int P, int32_t* leaf_of, uint64_t* leaf_key,
                                                               int32_t* leaf_start)
{
    const int s = blockIdx.x * PC_THREADS + threadIdx.x;
    if (s >= n) return;
    const int id = leaf_of[s] - 1;
    leaf_of[s] = id;
    if (s == 0 || keys[s] != keys[s - 1]) {
        leaf_key[id] = keys[s];
        leaf_start[id] = s;
    }
    if (s == n - 1) leaf_start[P] = n;
}

// ---- 3: sorted-order copy ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PC_THREADS) void pc_gather_kernel(const gpc_point_xyzrgb* cloud, const int32_t* vals, int n, PcPoint* sp)
{
    const int s = blockIdx.x * PC_THREADS + threadIdx.x;
    if (s >= n) return;
    const gpc_point_xyzrgb* q = &cloud[vals[s]];
    const float4 p = *reinterpret_cast<const float4*>(q);
    const uint32_t c = *reinterpret_cast<const uint32_t*>(&q->b);     // b | g << 8 | r << 16 | a << 24
    PcPoint o;
    o.x = p.x; o.y = p.y; o.z = p.z;
    o.rgb = ((c >> 16) & 0xffu) | (c & 0xff00u) | ((c & 0xffu) << 16);
    *reinterpret_cast<float4*>(&sp[s]) = *reinterpret_cast<const float4*>(&o);
}

// ---- 4: frames -------------------------------------------------------------------------------------------------------------
__device__ static inline int pc_find_leaf(const uint64_t* leaf_key, int P, uint64_t key)
{
    int lo = 0, hi = P;                       // first element >= key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (leaf_key[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < P && leaf_key[lo] == key) ? lo : -1;
}

__device__ static inline float pc_readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ static inline double pc_readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// eigenvector of the smallest eigenvalue of the symmetric 4x4 matrix A: the cyclic Jacobi of orc_smallest_eigvec4
__device__ static inline void pc_smallest_eigvec4(double A[4][4], double v[4])
{
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double offd = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) offd += A[p][q] * A[p][q];
        if (offd == 0.0) break;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double g = 100.0 * fabs(A[p][q]);       // negligible against both diagonal entries: drop it
                const bool drop = fabs(A[p][p]) + g == fabs(A[p][p]) && fabs(A[q][q]) + g == fabs(A[q][q]);
                if (A[p][q] != 0.0 && drop) A[p][q] = A[q][p] = 0.0;
                if (A[p][q] != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double akp = A[k][p], akq = A[k][q];
                        A[k][p] = c * akp - s * akq;
                        A[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double apk = A[p][k], aqk = A[q][k];
                        A[p][k] = c * apk - s * aqk;
                        A[q][k] = s * apk + c * aqk;
                    }
                    A[p][q] = A[q][p] = 0.0;                  // the rotation annihilates this pair: make it exact
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = c * vkp - s * vkq;
                        V[k][q] = s * vkp + c * vkq;
                    }
                }
            }
        }
    }
    double best = A[0][0];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = V[k][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (A[i][i] < best) {
            best = A[i][i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = V[k][i];
        }
}

__device__ static inline void pc_cross(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ static inline void pc_normalize(double a[3])
{
    const double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n > 0) { a[0] /= n; a[1] /= n; a[2] /= n; }
}

struct PcArgs {
    PcGrid g;
    int n, P;
    const uint64_t* leaf_key;
    const int32_t* leaf_start;
    const int32_t* leaf_of;
    const int32_t* vals;
    const PcPoint* sp;
    int32_t* nbr;          // P x 27 neighbour leaf ids (-1: empty voxel), (dz, dy, dx) order = ascending leaf order
    double* M;             // P x 16 moment matrices
    int32_t* kcount;       // P: points in the search sphere
    double* cen;           // P x 3 voxel centres
    int32_t* owner;        // per sorted position: owning leaf or -1
    double *py, *px0, *px1;   // per sorted position: patch-frame coordinates in the owner's frame
    int32_t* cnt;          // P + 1: points owned per leaf (cnt[P] = 0)
    int32_t* off;          // P + 1: exclusive scan of cnt
    int32_t* nmax;
    double *R, *mean, *rgb_mean;
    uint8_t* W;
    double *x0, *x1, *y, *rgb;
    int32_t* src;
};

#define PC_LROW 66         // LDS row pitch (doubles) of the product table: 64 hits + padding against bank conflicts

__global__ __launch_bounds__(PC_THREADS) void pc_moment_kernel(PcArgs A)
{
    // per wave: 10 rows (xx xy xz x yy yz y zz z 1) of the hits of one 64-point chunk, in hit order
    __shared__ double prod[PC_WAVES][10 * PC_LROW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int leaf = blockIdx.x * PC_WAVES + w;
    if (leaf >= A.P) return;                                  // whole waves leave; no block-level synchronisation below
    const PcGrid& g = A.g;
    double* pr = prod[w];
    int k3[3];
    pc_unpack(g, A.leaf_key[leaf], k3);
    double center[3];
    pc_center(g, k3, center);
    // the 27 neighbour segments: lane j looks up voxel (dz, dy, dx) = (j / 9, j / 3 % 3, j % 3) - 1
    int seg0 = 0, seg1 = 0;
    if (lane < 27) {
        const int nx = k3[0] + lane % 3 - 1, ny = k3[1] + (lane / 3) % 3 - 1, nz = k3[2] + lane / 9 - 1;
        int nb = -1;
        if (nx >= 0 && nx <= g.kmax[0] && ny >= 0 && ny <= g.kmax[1] && nz >= 0 && nz <= g.kmax[2])
            nb = pc_find_leaf(A.leaf_key, A.P, pc_pack(g, nx, ny, nz));
        A.nbr[(size_t)leaf * 27 + lane] = nb;
        if (nb >= 0) { seg0 = A.leaf_start[nb]; seg1 = A.leaf_start[nb + 1]; }
    }
    if (lane < 3) A.cen[(size_t)leaf * 3 + lane] = lane == 0 ? center[0] : (lane == 1 ? center[1] : center[2]);
    pr[9 * PC_LROW + lane] = 1.0;                             // the homogeneous coordinate's products
    const double r2 = g.radius * g.radius;
    // lanes 0..15: entry (ea, eb) of the moment matrix = product row of (min, max)
    const int ea = (lane >> 2) & 3, eb = lane & 3, lo = min(ea, eb), hi = max(ea, eb);
    const int row = (lo == 0 ? 0 : (lo == 1 ? 3 : (lo == 2 ? 5 : 6))) + hi;      // 0:0-3, 1:4-6, 2:7-8, 3:9
    const double* mine = pr + row * PC_LROW;
    double M = 0.0;
    int k = 0;
    for (int j = 0; j < 27; ++j) {
        const int s0 = __builtin_amdgcn_readlane(seg0, j), s1 = __builtin_amdgcn_readlane(seg1, j);
        for (int base = s0; base < s1; base += 64) {
            const int s = base + lane;
            double q0 = 0, q1 = 0, q2 = 0;
            bool in = false;
            if (s < s1) {
                const float4 p = *reinterpret_cast<const float4*>(&A.sp[s]);
                q0 = (double)p.x; q1 = (double)p.y; q2 = (double)p.z;
                const double ex = q0 - center[0], ey = q1 - center[1], ez = q2 - center[2];
                in = ex * ex + ey * ey + ez * ez <= r2;
            }
            const unsigned long long mask = __ballot(in);
            const int hits = __popcll(mask);
            if (in) {                                         // radiusSearch hit order = the oracle's accumulation order
                const int r = __popcll(mask & ((1ull << lane) - 1));
                pr[0 * PC_LROW + r] = q0 * q0; pr[1 * PC_LROW + r] = q0 * q1; pr[2 * PC_LROW + r] = q0 * q2; pr[3 * PC_LROW + r] = q0;
                pr[4 * PC_LROW + r] = q1 * q1; pr[5 * PC_LROW + r] = q1 * q2; pr[6 * PC_LROW + r] = q1;
                pr[7 * PC_LROW + r] = q2 * q2; pr[8 * PC_LROW + r] = q2;
            }
            __builtin_amdgcn_wave_barrier();                  // LDS is in order within a wave; keep the compiler in order too
#pragma unroll 8
            for (int r = 0; r < hits; ++r) M += mine[r];
            __builtin_amdgcn_wave_barrier();
            k += hits;
        }
    }
    if (lane < 16) A.M[(size_t)leaf * 16 + lane] = M;
    if (lane == 0) A.kcount[leaf] = k;
}

#define PC_FRAME_THREADS 64     // one wave per workgroup: the leaves spread over as many CUs as possible

__global__ __launch_bounds__(PC_FRAME_THREADS) void pc_frame_kernel(PcArgs A)
{
    const int leaf = blockIdx.x * PC_FRAME_THREADS + threadIdx.x;
    if (leaf >= A.P) return;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (A.kcount[leaf] >= 4) {                                // :31-34
        double Mm[4][4], v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) Mm[a][b] = A.M[(size_t)leaf * 16 + 4 * a + b];
        pc_smallest_eigvec4(Mm, v);
        double normal[3] = {v[0], v[1], v[2]};
        pc_normalize(normal);
        const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, z[3] = {0, 0, 1};
        double c1[3], c2[3];
        const double ax = fabs(normal[0]), ay = fabs(normal[1]), az = fabs(normal[2]);
        if (ax > ay && ax > az) {
            if (normal[0] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
            pc_cross(z, normal, c1);
        } else if (ay > ax && ay > az) {
            if (normal[1] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
            pc_cross(x, normal, c1);
        } else {
            if (normal[2] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
            pc_cross(y, normal, c1);
        }
        pc_normalize(c1);
        pc_cross(normal, c1, c2);
#pragma unroll
        for (int a = 0; a < 3; ++a) { R[a] = normal[a]; R[3 + a] = c1[a]; R[6 + a] = c2[a]; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) A.R[(size_t)leaf * 9 + i] = R[i];
}

// ---- 5: ownership ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PC_THREADS) void pc_claim_kernel(PcArgs A)
{
    __shared__ double frames[PC_WAVES][27][12];               // per wave: centre (3) + R (9) of the 27 candidate leaves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int leaf = blockIdx.x * PC_WAVES + w;
    if (leaf >= A.P) return;
    const PcGrid& g = A.g;
    int nb = -1;
    if (lane < 27) {
        nb = A.nbr[(size_t)leaf * 27 + lane];
        if (nb >= 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) frames[w][lane][i] = A.cen[(size_t)nb * 3 + i];
#pragma unroll
            for (int i = 0; i < 9; ++i) frames[w][lane][3 + i] = A.R[(size_t)nb * 9 + i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const double r2 = g.radius * g.radius;
    const int s0 = A.leaf_start[leaf], s1 = A.leaf_start[leaf + 1];
    for (int base = s0; base < s1; base += 64) {
        const int s = base + lane;
        const bool valid = s < s1;
        double p[3] = {0, 0, 0};
        if (valid) {
            const float4 f = *reinterpret_cast<const float4*>(&A.sp[s]);
            p[0] = (double)f.x; p[1] = (double)f.y; p[2] = (double)f.z;
        }
        int owner = -1;
        double pt[3] = {0, 0, 0};
        for (int j = 0; j < 27; ++j) {                        // candidates in leaf order: the first that accepts owns the point
            const int L = __builtin_amdgcn_readlane(nb, j);
            if (L < 0) continue;
            if (!__any(valid && owner < 0)) break;
            const double* fr = frames[w][j];
            const double d[3] = {p[0] - fr[0], p[1] - fr[1], p[2] - fr[2]};
            const bool in = valid && owner < 0 && d[0] * d[0] + d[1] * d[1] + d[2] * d[2] <= r2;    // in L's search sphere
            if (!__any(in)) continue;
            double q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) q[a] = fr[3 + 3 * a] * d[0] + fr[3 + 3 * a + 1] * d[1] + fr[3 + 3 * a + 2] * d[2];   // R^T d  (:84)
            const bool acc = in && !(q[1] > g.half || q[1] < -g.half || q[2] > g.half || q[2] < -g.half);     // :85-87
            if (acc) { owner = L; pt[0] = q[0]; pt[1] = q[1]; pt[2] = q[2]; }
            const int c = __popcll(__ballot(acc));
            if (c && lane == 0) atomicAdd(&A.cnt[L], c);
        }
        if (valid) {
            A.owner[s] = owner;
            A.py[s] = pt[0];
            A.px0[s] = pt[1];
            A.px1[s] = pt[2];
        }
    }
}

// nmax[0] = largest patch; nmax[1], nmax[2] = patches of <= 256 / <= 272 points: the size classes of the dense dispatch
// (gpc_api.hip), which the host reads together with n_max and hands to it so that the class launches are sized exactly
__global__ __launch_bounds__(PC_THREADS) void pc_nmax_kernel(const int32_t* cnt, int P, int32_t* nmax)
{
    int m = 0, c0 = 0, c1 = 0;
    for (int i = blockIdx.x * PC_THREADS + threadIdx.x; i < P; i += gridDim.x * PC_THREADS) {
        const int n = cnt[i];
        m = max(m, n);
        c0 += n <= 256;
        c1 += n <= 272;
    }
    for (int o = 32; o > 0; o >>= 1) {
        m = max(m, __shfl_xor(m, o));
        c0 += __shfl_xor(c0, o);
        c1 += __shfl_xor(c1, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(nmax, m);
        atomicAdd(nmax + 1, c0);
        atomicAdd(nmax + 2, c1);
    }
}

// ---- 6: ordered compaction, means, mask ------------------------------------------------------------------------------------
__global__ __launch_bounds__(PC_THREADS) void pc_emit_kernel(PcArgs A)
{
    const int lane = threadIdx.x & 63;
    const int leaf = blockIdx.x * PC_WAVES + (threadIdx.x >> 6);
    if (leaf >= A.P) return;
    const PcGrid& g = A.g;
    const int base_q = A.off[leaf], cnt = A.off[leaf + 1] - base_q, total = A.off[A.P];
    int seg0 = 0, seg1 = 0;
    if (lane < 27) {
        const int nb = A.nbr[(size_t)leaf * 27 + lane];
        if (nb >= 0) { seg0 = A.leaf_start[nb]; seg1 = A.leaf_start[nb + 1]; }
    }
    int k3[3];
    pc_unpack(g, A.leaf_key[leaf], k3);
    double mid[3];
    pc_center(g, k3, mid);
    // pass 1: depth sum in patch order (the oracle's sequential sum), colour sums (integers: any order)
    double mnd = 0.0;
    int cs[3] = {0, 0, 0};
    for (int j = 0; j < 27; ++j) {
        const int s0 = __builtin_amdgcn_readlane(seg0, j), s1 = __builtin_amdgcn_readlane(seg1, j);
        for (int b0 = s0; b0 < s1; b0 += 64) {
            const int s = b0 + lane;
            const bool mine = s < s1 && A.owner[s] == leaf;
            double d = 0.0;
            if (mine) {
                d = A.py[s];
                const uint32_t c = A.sp[s].rgb;
                cs[0] += (int)(c & 0xffu); cs[1] += (int)((c >> 8) & 0xffu); cs[2] += (int)((c >> 16) & 0xffu);
            }
            unsigned long long mask = __ballot(mine);
            while (mask) {
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                mnd += pc_readlane_d(d, b);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) cs[a] += __shfl_xor(cs[a], o);
    double cmean[3] = {0, 0, 0};
    if (cnt > 0) {                                                           // :101-107, :116
        mnd /= (double)cnt;
#pragma unroll
        for (int a = 0; a < 3; ++a) cmean[a] = (double)cs[a] / (double)cnt;
        const double* R = A.R + (size_t)leaf * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) mid[a] += mnd * R[a];
    }
    if (lane < 3) {
        A.mean[(size_t)leaf * 3 + lane] = lane == 0 ? mid[0] : (lane == 1 ? mid[1] : mid[2]);
        A.rgb_mean[(size_t)leaf * 3 + lane] = lane == 0 ? cmean[0] : (lane == 1 ? cmean[1] : cmean[2]);
    }
    // pass 2: the same walk, writing the patch in hit order
    uint8_t* W = A.W + (size_t)leaf * (size_t)(g.sz * g.sz);
    int run = 0;
    for (int j = 0; j < 27; ++j) {
        const int s0 = __builtin_amdgcn_readlane(seg0, j), s1 = __builtin_amdgcn_readlane(seg1, j);
        for (int b0 = s0; b0 < s1; b0 += 64) {
            const int s = b0 + lane;
            const bool mine = s < s1 && A.owner[s] == leaf;
            const unsigned long long mask = __ballot(mine);
            if (mine) {
                const size_t q = (size_t)base_q + run + __popcll(mask & ((1ull << lane) - 1));
                const double u = A.px0[s], w = A.px1[s];
                A.y[q] = A.py[s] - mnd;
                A.x0[q] = u;
                A.x1[q] = w;
                A.src[q] = A.vals[s];
                const uint32_t c = A.sp[s].rgb;
                A.rgb[q] = (double)(c & 0xffu) - cmean[0];
                A.rgb[(size_t)total + q] = (double)((c >> 8) & 0xffu) - cmean[1];
                A.rgb[2 * (size_t)total + q] = (double)((c >> 16) & 0xffu) - cmean[2];
                int gx = (int)((double)g.sz * (u / g.res + 0.5)), gy = (int)((double)g.sz * (w / g.res + 0.5));   // :90-92
                gx = min(max(gx, 0), g.sz - 1);
                gy = min(max(gy, 0), g.sz - 1);
                W[g.sz * gx + gy] = 1;
            }
            run += __popcll(mask);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct gpc_patches {
    gpc_ctx* ctx = nullptr;
    gpc_patches_view v{};       // device pointers into `block`
    void* block = nullptr;      // one allocation holds every array of the batch
};

namespace {

int bits_for(int kmax)
{
    int b = 1;
    while ((1ll << b) <= (long long)kmax) ++b;
    return b;
}

// carves 256-byte aligned pieces out of one buffer; with base == nullptr it only measures
struct Carver {
    char* base;
    size_t used = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <class T> T* take(size_t count)
    {
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

struct Scratch {                // per call, in the context's workspace
    uint32_t* bounds;
    uint64_t *k0, *k1, *leaf_key;
    int32_t *v0, *vals, *head, *leaf_of, *leaf_start, *nbr, *kcount, *owner, *cnt;
    PcPoint* sp;
    double *py, *px0, *px1, *M, *cen;
    void* prim;
    size_t prim_bytes;
};

size_t carve_scratch(Carver& c, Scratch& s, size_t n, size_t pb, size_t prim_bytes)
{
    s.bounds = c.take<uint32_t>(8);
    s.k0 = c.take<uint64_t>(n); s.k1 = c.take<uint64_t>(n);
    s.v0 = c.take<int32_t>(n); s.vals = c.take<int32_t>(n);
    s.head = c.take<int32_t>(n); s.leaf_of = c.take<int32_t>(n);
    s.owner = c.take<int32_t>(n);
    s.sp = c.take<PcPoint>(n);
    s.py = c.take<double>(n); s.px0 = c.take<double>(n); s.px1 = c.take<double>(n);
    s.leaf_key = c.take<uint64_t>(pb);
    s.leaf_start = c.take<int32_t>(pb + 1);
    s.nbr = c.take<int32_t>(27 * pb);
    s.kcount = c.take<int32_t>(pb);
    s.cnt = c.take<int32_t>(pb + 4);      // P + 1 counts | n_max | patches of <= 256 points | patches of <= 272 points
    s.M = c.take<double>(16 * pb);
    s.cen = c.take<double>(3 * pb);
    s.prim = c.take<char>(prim_bytes);
    s.prim_bytes = prim_bytes;
    return c.used;
}

}  // namespace

#define PC_HIP(call)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            pc_patches_release(o);                                                                            \
            return gpc_fail(ctx, e_ == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_project_cloud: %s failed: %s", \
                            #call, hipGetErrorString(e_));                                                     \
        }                                                                                                      \
    } while (0)

extern "C" {

// Safe in either order with gpc_ctx_destroy (the batch holds a reference on its context; hipFree synchronises the device).
// the caller holds ctx->mu (or the object was never published): the error paths of gpc_project_cloud_dev end here
static void pc_patches_release(gpc_patches* o)
{
    if (!o) return;
    gpc_ctx* ctx = o->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (ctx && ctx->hint_off == o->v.off) ctx->hint_off = nullptr;
    if (o->block) (void)hipFree(o->block);
    delete o;
    if (ctx) gpc_ctx_unref(ctx);
}

void gpc_patches_destroy(gpc_patches* o)
{
    if (!o) return;
    gpc_ctx* ctx = o->ctx;
    if (!ctx) { pc_patches_release(o); return; }
    gpc_ctx_ref(ctx);                                  // the release may drop the last reference while the lock is held
    {
        std::lock_guard<std::mutex> lk(ctx->mu);       // dense_dispatch reads the size-class hint under the same lock
        pc_patches_release(o);
    }
    gpc_ctx_unref(ctx);
}

int gpc_project_cloud_dev(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "negative point count");
    if (n > 0 && !cloud) return gpc_fail(ctx, GPC_EINVAL, "cloud is NULL");
    if (!(res > 0.0) || !(res < 1e300)) return gpc_fail(ctx, GPC_EINVAL, "res must be positive and finite");
    if (sz < 1 || sz > 1024) return gpc_fail(ctx, GPC_EINVAL, "sz must be in [1, 1024]");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    hipStream_t st = ctx->stream;
    gpc_patches* o = new gpc_patches;
    o->ctx = ctx;
    gpc_ctx_ref(ctx);
    o->v.m = sz * sz;
    if (n == 0) {
        PC_HIP(hipMalloc(&o->block, 256));
        PC_HIP(hipMemsetAsync(o->block, 0, 256, st));
        PC_HIP(hipStreamSynchronize(st));
        o->v.off = static_cast<int32_t*>(o->block);
        *out = o;
        return GPC_OK;
    }
    const int nblk = (n + PC_THREADS - 1) / PC_THREADS;
    const size_t N = (size_t)n;

    // 1: bounds (its 32 bytes of scratch sit at the start of the workspace whatever the leaf bound turns out to be)
    {
        const int rc = gpc_ws_reserve(ctx, 4096);
        if (rc != GPC_OK) { pc_patches_release(o); return rc; }
    }
    uint32_t* d_bounds = static_cast<uint32_t*>(ctx->ws);
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0};
    PC_HIP(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pc_bounds_kernel, dim3(nblk < ctx->num_cus * 4 ? nblk : ctx->num_cus * 4), dim3(PC_THREADS), 0, st, cloud, n, d_bounds);
    PC_HIP(hipGetLastError());
    uint32_t hb[8];
    PC_HIP(hipMemcpyAsync(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    if (hb[6]) {
        pc_patches_release(o);
        return gpc_fail(ctx, GPC_EINVAL, "the cloud holds a non-finite coordinate");
    }
    PcGrid g;
    g.res = res;
    g.radius = std::sqrt(3.0f) / 2.0f * res;            // :194
    g.half = res / 2.0f;
    g.sz = sz;
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) {
        g.mn[a] = (double)pc_unordered(hb[a]);
        const double ext = std::floor(((double)pc_unordered(hb[3 + a]) - g.mn[a]) / res);
        if (!(ext < 2097152.0)) {
            pc_patches_release(o);
            return gpc_fail(ctx, GPC_ERANGE, "more than 2^21 voxels of side res along an axis");
        }
        g.kmax[a] = (int)ext;
        cells *= ext + 1.0;
    }
    g.bx = bits_for(g.kmax[0]); g.by = bits_for(g.kmax[1]); g.bz = bits_for(g.kmax[2]);
    const int key_bits = g.bx + g.by + g.bz;           // <= 63
    const size_t pb = cells < (double)n ? (size_t)cells : N;   // bound on the number of leaves

    // scratch
    size_t sort_bytes = 0, scan_bytes = 0, scan2_bytes = 0;
    PC_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, N,
                                     0u, (unsigned)key_bits, st));
    PC_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, N, rocprim::plus<int32_t>(), st));
    PC_HIP(rocprim::exclusive_scan(nullptr, scan2_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, pb + 1, rocprim::plus<int32_t>(), st));
    const size_t prim_bytes = std::max(sort_bytes, std::max(scan_bytes, scan2_bytes));
    Scratch S;
    {
        Carver measure(nullptr);
        const int rc = gpc_ws_reserve(ctx, carve_scratch(measure, S, N, pb, prim_bytes));
        if (rc != GPC_OK) { pc_patches_release(o); return rc; }
        Carver c(ctx->ws);
        carve_scratch(c, S, N, pb, prim_bytes);
    }

    // 2: keys, stable sort, leaf table
    hipLaunchKernelGGL(pc_keys_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, g, cloud, n, S.k0, S.v0);
    PC_HIP(hipGetLastError());
    size_t tb = S.prim_bytes;
    PC_HIP(rocprim::radix_sort_pairs(S.prim, tb, S.k0, S.k1, S.v0, S.vals, N, 0u, (unsigned)key_bits, st));
    hipLaunchKernelGGL(pc_heads_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, S.k1, n, S.head);
    PC_HIP(hipGetLastError());
    tb = S.prim_bytes;
    PC_HIP(rocprim::inclusive_scan(S.prim, tb, S.head, S.leaf_of, N, rocprim::plus<int32_t>(), st));
    int32_t P = 0;
    PC_HIP(hipMemcpyAsync(&P, S.leaf_of + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // 3: sorted-order copy (does not need P: overlaps the read-back)
    hipLaunchKernelGGL(pc_gather_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, cloud, S.vals, n, S.sp);
    PC_HIP(hipGetLastError());
    PC_HIP(hipStreamSynchronize(st));
    if ((long long)P * (long long)(sz * sz) > 0x7fffffffLL) {
        pc_patches_release(o);
        return gpc_fail(ctx, GPC_ERANGE, "P * sz * sz exceeds 2^31-1");
    }
    if ((size_t)P > pb) {
        pc_patches_release(o);
        return gpc_fail(ctx, GPC_EHIP, "internal: %d leaves exceed the bound %zu", (int)P, pb);
    }

    // the result: one block; per-point arrays are sized by n (an upper bound of the points owned)
    const size_t Pz = (size_t)P, m = (size_t)(sz * sz);
    Carver oc(nullptr);
    for (int pass = 0; pass < 2; ++pass) {
        oc = Carver(pass ? o->block : nullptr);
        o->v.off = oc.take<int32_t>(Pz + 1);
        o->v.rotations = oc.take<double>(9 * Pz);
        o->v.means = oc.take<double>(3 * Pz);
        o->v.rgb_means = oc.take<double>(3 * Pz);
        o->v.W = oc.take<uint8_t>(Pz * m);
        o->v.x0 = oc.take<double>(N);
        o->v.x1 = oc.take<double>(N);
        o->v.y = oc.take<double>(N);
        o->v.rgb = oc.take<double>(3 * N);
        o->v.src = oc.take<int32_t>(N);
        if (!pass) PC_HIP(hipMalloc(&o->block, oc.used));
    }
    PcArgs A;
    memset(&A, 0, sizeof(A));
    A.g = g; A.n = n; A.P = P;
    A.leaf_key = S.leaf_key; A.leaf_start = S.leaf_start; A.leaf_of = S.leaf_of; A.vals = S.vals; A.sp = S.sp;
    A.nbr = S.nbr; A.M = S.M; A.kcount = S.kcount; A.cen = S.cen; A.owner = S.owner; A.py = S.py; A.px0 = S.px0; A.px1 = S.px1;
    A.cnt = S.cnt; A.nmax = S.cnt + (P + 1);
    A.off = const_cast<int32_t*>(o->v.off); A.R = const_cast<double*>(o->v.rotations); A.mean = const_cast<double*>(o->v.means);
    A.rgb_mean = const_cast<double*>(o->v.rgb_means); A.W = const_cast<uint8_t*>(o->v.W);
    A.x0 = const_cast<double*>(o->v.x0); A.x1 = const_cast<double*>(o->v.x1); A.y = const_cast<double*>(o->v.y);
    A.rgb = const_cast<double*>(o->v.rgb); A.src = const_cast<int32_t*>(o->v.src);
    PC_HIP(hipMemsetAsync(S.cnt, 0, sizeof(int32_t) * (Pz + 4), st));
    PC_HIP(hipMemsetAsync(A.W, 0, Pz * m, st));
    hipLaunchKernelGGL(pc_leaves_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, S.k1, n, (int)P, S.leaf_of, S.leaf_key, S.leaf_start);
    PC_HIP(hipGetLastError());
    // 4: frames
    const int lblk = (P + PC_WAVES - 1) / PC_WAVES;
    hipLaunchKernelGGL(pc_moment_kernel, dim3(lblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    hipLaunchKernelGGL(pc_frame_kernel, dim3((P + PC_FRAME_THREADS - 1) / PC_FRAME_THREADS), dim3(PC_FRAME_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    // 5: ownership, offsets
    hipLaunchKernelGGL(pc_claim_kernel, dim3(lblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    tb = S.prim_bytes;
    PC_HIP(rocprim::exclusive_scan(S.prim, tb, S.cnt, A.off, (int32_t)0, Pz + 1, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(pc_nmax_kernel, dim3(64), dim3(PC_THREADS), 0, st, S.cnt, (int)P, A.nmax);
    PC_HIP(hipGetLastError());
    // 6: the patch batch (the colour planes' pitch is the total, read from off[P] on the device)
    hipLaunchKernelGGL(pc_emit_kernel, dim3(lblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    int32_t total = 0, nmax[3] = {0, 0, 0};
    PC_HIP(hipMemcpyAsync(&total, A.off + P, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipMemcpyAsync(nmax, A.nmax, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    o->v.P = P; o->v.n_total = total; o->v.n_max = nmax[0];
    // the size classes of this batch, for the dense dispatch (keyed by the batch's own `off` buffer, which lives as long as the object)
    ctx->hint_off = o->v.off; ctx->hint_P = P; ctx->hint_le256 = nmax[1]; ctx->hint_le272 = nmax[2];
    *out = o;
    return GPC_OK;
}

int gpc_project_cloud(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "negative point count");
    if (n > 0 && !cloud) return gpc_fail(ctx, GPC_EINVAL, "cloud is NULL");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void* d_cloud = nullptr;
    if (n > 0) {
        GPC_HIP(ctx, hipMalloc(&d_cloud, sizeof(gpc_point_xyzrgb) * (size_t)n));
        hipError_t e = hipMemcpyAsync(d_cloud, cloud, sizeof(gpc_point_xyzrgb) * (size_t)n, hipMemcpyHostToDevice, gpc_stream_of(ctx));
        if (e != hipSuccess) {
            (void)hipFree(d_cloud);
            return gpc_fail(ctx, GPC_EHIP, "gpc_project_cloud: upload failed: %s", hipGetErrorString(e));
        }
    }
    const int rc = gpc_project_cloud_dev(ctx, (const gpc_point_xyzrgb*)d_cloud, n, res, sz, out);
    if (d_cloud) {
        (void)hipStreamSynchronize(gpc_stream_of(ctx));
        (void)hipFree(d_cloud);
    }
    return rc;
}

int gpc_patches_view_dev(const gpc_patches* p, gpc_patches_view* view)
{
    if (!p || !view) return GPC_EINVAL;
    *view = p->v;
    return GPC_OK;
}

int gpc_patches_fetch(const gpc_patches* p, int32_t* off, double* x0, double* x1, double* y, double* rgb, double* rotations,
                      double* means, double* rgb_means, uint8_t* W, int32_t* src)
{
    if (!p) return GPC_EINVAL;
    gpc_ctx* ctx = p->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the batch can only be destroyed
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const gpc_patches_view& v = p->v;
    const size_t P = (size_t)v.P, N = (size_t)v.n_total;
    struct { void* dst; const void* src; size_t bytes; } cp[10] = {
        {off, v.off, 4 * (P + 1)}, {x0, v.x0, 8 * N}, {x1, v.x1, 8 * N}, {y, v.y, 8 * N}, {rgb, v.rgb, 24 * N},
        {rotations, v.rotations, 72 * P}, {means, v.means, 24 * P}, {rgb_means, v.rgb_means, 24 * P}, {W, v.W, P * (size_t)v.m},
        {src, v.src, 4 * N}};
    hipStream_t s = gpc_stream_of(ctx);
    for (auto& c : cp)
        if (c.dst && c.bytes) GPC_HIP(ctx, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

}  // extern "C"
