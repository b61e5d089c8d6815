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
//   1 pc_bounds_kernel    min / max corner, finiteness                                  one pass over the cloud
//   2 pc_keys_kernel      voxel key (z, y, x packed) per point, then rocPRIM's stable radix sort of (key, index) over
//                         exactly the key's bits, pc_heads/pc_leaves: leaf table (sorted unique keys + segment starts)
//   3 pc_gather_kernel    points re-laid in sorted order as 16-byte records (x, y, z, rgb): every later access to a
//                         voxel's points is one contiguous, coalesced segment
//   4 pc_rotation_kernel  one wave per leaf: 27 neighbour segments (binary search in the leaf table), sphere test by
//                         ballot, 4x4 moment matrix accumulated in the oracle's order (16 lanes, one entry each),
//                         cyclic Jacobi eigen-solve, frame construction (:37-63)
//   5 pc_claim_kernel     one thread per point: first accepting leaf among the 27 neighbours; patch-frame coordinates
//   6 pc_emit_kernel      one wave per leaf: ordered compaction (ballot + prefix popcount) of the points it owns, depth
//                         mean as the oracle's sequential sum, colour means, mean removal, centre shift, mask W
// Bit-exactness: floating-point contraction is off for this file and every expression is written in the association of
// oracle/gpc_oracle_producer.c; products of two floats are exact in double, so the moment sums only fix the ORDER.
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
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&out[a], lo[a]);
            atomicMax(&out[3 + a], hi[a]);
        }
        if (bad) atomicOr(&out[6], 1u);
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
        if (offd < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                if (!(fabs(A[p][q]) < 1e-300)) {
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

__global__ __launch_bounds__(PC_THREADS) void pc_rotation_kernel(PcArgs A)
{
    const int lane = threadIdx.x & 63;
    const int leaf = blockIdx.x * PC_WAVES + (threadIdx.x >> 6);
    if (leaf >= A.P) return;                                  // whole waves leave; no block-level synchronisation below
    const PcGrid& g = A.g;
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
    const double r2 = g.radius * g.radius;
    const int ea = (lane >> 2) & 3, eb = lane & 3;            // lanes 0..15: entry (ea, eb) of the moment matrix
    double M = 0.0;
    int k = 0;
    for (int j = 0; j < 27; ++j) {
        const int s0 = __builtin_amdgcn_readlane(seg0, j), s1 = __builtin_amdgcn_readlane(seg1, j);
        for (int base = s0; base < s1; base += 64) {
            const int s = base + lane;
            float px = 0.f, py = 0.f, pz = 0.f;
            bool in = false;
            if (s < s1) {
                const float4 p = *reinterpret_cast<const float4*>(&A.sp[s]);
                px = p.x; py = p.y; pz = p.z;
                const double ex = (double)px - center[0], ey = (double)py - center[1], ez = (double)pz - center[2];
                in = ex * ex + ey * ey + ez * ez <= r2;
            }
            unsigned long long mask = __ballot(in);
            k += __popcll(mask);
            while (mask) {                                    // radiusSearch hit order = the oracle's accumulation order
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                const double q0 = (double)pc_readlane_f(px, b), q1 = (double)pc_readlane_f(py, b), q2 = (double)pc_readlane_f(pz, b);
                const double va = ea == 0 ? q0 : (ea == 1 ? q1 : (ea == 2 ? q2 : 1.0));
                const double vb = eb == 0 ? q0 : (eb == 1 ? q1 : (eb == 2 ? q2 : 1.0));
                M += va * vb;
            }
        }
    }
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (k >= 4) {                                             // :31-34
        double Mm[4][4], v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) Mm[a][b] = pc_readlane_d(M, 4 * a + b);
        pc_smallest_eigvec4(Mm, v);                          // wave-uniform: every lane solves the same 4x4 problem
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
    if (lane < 9) {
        double r = R[0];
#pragma unroll
        for (int i = 1; i < 9; ++i) r = lane == i ? R[i] : r;
        A.R[(size_t)leaf * 9 + lane] = r;
    }
}

// ---- 5: ownership ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PC_THREADS) void pc_claim_kernel(PcArgs A)
{
    const int s = blockIdx.x * PC_THREADS + threadIdx.x;
    if (s >= A.n) return;
    const PcGrid& g = A.g;
    const float4 p = *reinterpret_cast<const float4*>(&A.sp[s]);
    const int32_t* nbr = A.nbr + (size_t)A.leaf_of[s] * 27;
    const double r2 = g.radius * g.radius;
    int owner = -1;
    double pt[3] = {0, 0, 0};
    for (int j = 0; j < 27 && owner < 0; ++j) {
        const int L = nbr[j];
        if (L < 0) continue;
        int k3[3];
        pc_unpack(g, A.leaf_key[L], k3);
        double c[3];
        pc_center(g, k3, c);
        const double d[3] = {(double)p.x - c[0], (double)p.y - c[1], (double)p.z - c[2]};
        if (!(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] <= r2)) continue;          // not in this leaf's search sphere
        const double* R = A.R + (size_t)L * 9;
        double q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = R[3 * a] * d[0] + R[3 * a + 1] * d[1] + R[3 * a + 2] * d[2];   // R^T d  (:84)
        if (q[1] > g.half || q[1] < -g.half || q[2] > g.half || q[2] < -g.half) continue;                // :85-87
        owner = L;
        pt[0] = q[0]; pt[1] = q[1]; pt[2] = q[2];
    }
    A.owner[s] = owner;
    A.py[s] = pt[0];
    A.px0[s] = pt[1];
    A.px1[s] = pt[2];
    if (owner >= 0) atomicAdd(&A.cnt[owner], 1);
}

__global__ __launch_bounds__(PC_THREADS) void pc_nmax_kernel(const int32_t* cnt, int P, int32_t* nmax)
{
    int m = 0;
    for (int i = blockIdx.x * PC_THREADS + threadIdx.x; i < P; i += gridDim.x * PC_THREADS) m = max(m, cnt[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(nmax, m);
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
    gpc_patches_view v{};       // device pointers
    void* bufs[12] = {};
    int nbufs = 0;
};

namespace {

struct DevBufs {                // scratch that dies with the call
    void* p[24] = {};
    int n = 0;
    hipError_t get(void** out, size_t bytes)
    {
        hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) p[n++] = *out;
        return e;
    }
    ~DevBufs()
    {
        for (int i = 0; i < n; ++i) (void)hipFree(p[i]);
    }
};

int bits_for(int kmax)
{
    int b = 1;
    while ((1ll << b) <= (long long)kmax) ++b;
    return b;
}

hipError_t keep(gpc_patches* o, void** out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e == hipSuccess) o->bufs[o->nbufs++] = *out;
    return e;
}

}  // namespace

#define PC_HIP(call)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            gpc_patches_destroy(o);                                                                            \
            return gpc_fail(ctx, e_ == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_project_cloud: %s failed: %s", \
                            #call, hipGetErrorString(e_));                                                     \
        }                                                                                                      \
    } while (0)

extern "C" {

void gpc_patches_destroy(gpc_patches* o)
{
    if (!o) return;
    for (int i = 0; i < o->nbufs; ++i) (void)hipFree(o->bufs[i]);
    delete o;
}

int gpc_project_cloud_dev(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out)
{
    if (!ctx) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "negative point count");
    if (n > 0 && !cloud) return gpc_fail(ctx, GPC_EINVAL, "cloud is NULL");
    if (!(res > 0.0) || !(res < 1e300)) return gpc_fail(ctx, GPC_EINVAL, "res must be positive and finite");
    if (sz < 1 || sz > 1024) return gpc_fail(ctx, GPC_EINVAL, "sz must be in [1, 1024]");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    gpc_patches* o = new gpc_patches;
    o->ctx = ctx;
    o->v.m = sz * sz;
    int32_t* d_off = nullptr;
    if (n == 0) {
        PC_HIP(keep(o, (void**)&d_off, sizeof(int32_t)));
        PC_HIP(hipMemsetAsync(d_off, 0, sizeof(int32_t), st));
        PC_HIP(hipStreamSynchronize(st));
        o->v.off = d_off;
        *out = o;
        return GPC_OK;
    }
    DevBufs tmp;
    const int nblk = (n + PC_THREADS - 1) / PC_THREADS;

    // 1: bounds
    uint32_t* d_bounds = nullptr;
    PC_HIP(tmp.get((void**)&d_bounds, 8 * sizeof(uint32_t)));
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0};
    PC_HIP(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pc_bounds_kernel, dim3(nblk < ctx->num_cus * 8 ? nblk : ctx->num_cus * 8), dim3(PC_THREADS), 0, st, cloud, n, d_bounds);
    PC_HIP(hipGetLastError());
    uint32_t hb[8];
    PC_HIP(hipMemcpyAsync(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    if (hb[6]) {
        gpc_patches_destroy(o);
        return gpc_fail(ctx, GPC_EINVAL, "the cloud holds a non-finite coordinate");
    }
    PcGrid g;
    g.res = res;
    g.radius = std::sqrt(3.0f) / 2.0f * res;            // :194
    g.half = res / 2.0f;
    g.sz = sz;
    for (int a = 0; a < 3; ++a) {
        g.mn[a] = (double)pc_unordered(hb[a]);
        const double ext = std::floor(((double)pc_unordered(hb[3 + a]) - g.mn[a]) / res);
        if (!(ext < 2097152.0)) {
            gpc_patches_destroy(o);
            return gpc_fail(ctx, GPC_ERANGE, "more than 2^21 voxels of side res along an axis");
        }
        g.kmax[a] = (int)ext;
    }
    g.bx = bits_for(g.kmax[0]); g.by = bits_for(g.kmax[1]); g.bz = bits_for(g.kmax[2]);
    const int key_bits = g.bx + g.by + g.bz;           // <= 63

    // 2: keys, stable sort, leaf table
    uint64_t *d_k0 = nullptr, *d_k1 = nullptr;
    int32_t *d_v0 = nullptr, *d_vals = nullptr, *d_head = nullptr, *d_leaf_of = nullptr;
    PC_HIP(tmp.get((void**)&d_k0, sizeof(uint64_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_k1, sizeof(uint64_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_v0, sizeof(int32_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_vals, sizeof(int32_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_head, sizeof(int32_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_leaf_of, sizeof(int32_t) * (size_t)n));
    hipLaunchKernelGGL(pc_keys_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, g, cloud, n, d_k0, d_v0);
    PC_HIP(hipGetLastError());
    size_t sort_bytes = 0, scan_bytes = 0, scan2_bytes = 0;
    PC_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_k0, d_k1, d_v0, d_vals, (size_t)n, 0u, (unsigned)key_bits, st));
    PC_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, d_head, d_leaf_of, (size_t)n, rocprim::plus<int32_t>(), st));
    void* d_tmp = nullptr;
    PC_HIP(tmp.get(&d_tmp, sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    PC_HIP(rocprim::radix_sort_pairs(d_tmp, sort_bytes, d_k0, d_k1, d_v0, d_vals, (size_t)n, 0u, (unsigned)key_bits, st));
    hipLaunchKernelGGL(pc_heads_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, d_k1, n, d_head);
    PC_HIP(hipGetLastError());
    PC_HIP(rocprim::inclusive_scan(d_tmp, scan_bytes, d_head, d_leaf_of, (size_t)n, rocprim::plus<int32_t>(), st));
    int32_t P = 0;
    PC_HIP(hipMemcpyAsync(&P, d_leaf_of + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    if ((long long)P * (long long)(sz * sz) > 0x7fffffffLL) {
        gpc_patches_destroy(o);
        return gpc_fail(ctx, GPC_ERANGE, "P * sz * sz exceeds 2^31-1");
    }

    PcArgs A;
    memset(&A, 0, sizeof(A));
    A.g = g; A.n = n; A.P = P;
    uint64_t* d_leaf_key = nullptr;
    int32_t *d_leaf_start = nullptr, *d_nbr = nullptr, *d_owner = nullptr, *d_cnt = nullptr, *d_nmax = nullptr;
    PcPoint* d_sp = nullptr;
    double *d_py = nullptr, *d_px0 = nullptr, *d_px1 = nullptr;
    PC_HIP(tmp.get((void**)&d_leaf_key, sizeof(uint64_t) * (size_t)P));
    PC_HIP(tmp.get((void**)&d_leaf_start, sizeof(int32_t) * ((size_t)P + 1)));
    PC_HIP(tmp.get((void**)&d_nbr, sizeof(int32_t) * 27 * (size_t)P));
    PC_HIP(tmp.get((void**)&d_owner, sizeof(int32_t) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_cnt, sizeof(int32_t) * ((size_t)P + 2)));
    PC_HIP(tmp.get((void**)&d_sp, sizeof(PcPoint) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_py, sizeof(double) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_px0, sizeof(double) * (size_t)n));
    PC_HIP(tmp.get((void**)&d_px1, sizeof(double) * (size_t)n));
    d_nmax = d_cnt + (P + 1);
    double *d_R = nullptr, *d_mean = nullptr, *d_cmean = nullptr;
    uint8_t* d_W = nullptr;
    PC_HIP(keep(o, (void**)&d_off, sizeof(int32_t) * ((size_t)P + 1)));
    PC_HIP(keep(o, (void**)&d_R, sizeof(double) * 9 * (size_t)P));
    PC_HIP(keep(o, (void**)&d_mean, sizeof(double) * 3 * (size_t)P));
    PC_HIP(keep(o, (void**)&d_cmean, sizeof(double) * 3 * (size_t)P));
    PC_HIP(keep(o, (void**)&d_W, (size_t)P * (size_t)(sz * sz)));
    PC_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * ((size_t)P + 2), st));
    PC_HIP(hipMemsetAsync(d_W, 0, (size_t)P * (size_t)(sz * sz), st));
    A.leaf_key = d_leaf_key; A.leaf_start = d_leaf_start; A.leaf_of = d_leaf_of; A.vals = d_vals; A.sp = d_sp;
    A.nbr = d_nbr; A.owner = d_owner; A.py = d_py; A.px0 = d_px0; A.px1 = d_px1; A.cnt = d_cnt; A.off = d_off; A.nmax = d_nmax;
    A.R = d_R; A.mean = d_mean; A.rgb_mean = d_cmean; A.W = d_W;
    hipLaunchKernelGGL(pc_leaves_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, d_k1, n, (int)P, d_leaf_of, d_leaf_key, d_leaf_start);
    PC_HIP(hipGetLastError());
    // 3: sorted-order copy
    hipLaunchKernelGGL(pc_gather_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, cloud, d_vals, n, d_sp);
    PC_HIP(hipGetLastError());
    // 4: frames
    const int lblk = (P + PC_WAVES - 1) / PC_WAVES;
    hipLaunchKernelGGL(pc_rotation_kernel, dim3(lblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    // 5: ownership, offsets
    hipLaunchKernelGGL(pc_claim_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    PC_HIP(rocprim::exclusive_scan(nullptr, scan2_bytes, d_cnt, d_off, (int32_t)0, (size_t)P + 1, rocprim::plus<int32_t>(), st));
    void* d_tmp2 = nullptr;
    PC_HIP(tmp.get(&d_tmp2, scan2_bytes));
    PC_HIP(rocprim::exclusive_scan(d_tmp2, scan2_bytes, d_cnt, d_off, (int32_t)0, (size_t)P + 1, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(pc_nmax_kernel, dim3(64), dim3(PC_THREADS), 0, st, d_cnt, (int)P, d_nmax);
    PC_HIP(hipGetLastError());
    int32_t total = 0, nmax = 0;
    PC_HIP(hipMemcpyAsync(&total, d_off + P, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipMemcpyAsync(&nmax, d_nmax, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    // 6: the patch batch
    double *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_rgb = nullptr;
    int32_t* d_src = nullptr;
    PC_HIP(keep(o, (void**)&d_x0, sizeof(double) * (size_t)total));
    PC_HIP(keep(o, (void**)&d_x1, sizeof(double) * (size_t)total));
    PC_HIP(keep(o, (void**)&d_y, sizeof(double) * (size_t)total));
    PC_HIP(keep(o, (void**)&d_rgb, sizeof(double) * 3 * (size_t)total));
    PC_HIP(keep(o, (void**)&d_src, sizeof(int32_t) * (size_t)total));
    A.x0 = d_x0; A.x1 = d_x1; A.y = d_y; A.rgb = d_rgb; A.src = d_src;
    hipLaunchKernelGGL(pc_emit_kernel, dim3(lblk), dim3(PC_THREADS), 0, st, A);
    PC_HIP(hipGetLastError());
    PC_HIP(hipStreamSynchronize(st));                  // the scratch buffers are freed on return
    o->v.P = P; o->v.n_total = total; o->v.n_max = nmax;
    o->v.off = d_off; o->v.x0 = d_x0; o->v.x1 = d_x1; o->v.y = d_y; o->v.rgb = d_rgb; o->v.rotations = d_R; o->v.means = d_mean;
    o->v.rgb_means = d_cmean; o->v.W = d_W; o->v.src = d_src;
    *out = o;
    return GPC_OK;
}

int gpc_project_cloud(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out)
{
    if (!ctx) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "negative point count");
    if (n > 0 && !cloud) return gpc_fail(ctx, GPC_EINVAL, "cloud is NULL");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void* d_cloud = nullptr;
    if (n > 0) {
        GPC_HIP(ctx, hipMalloc(&d_cloud, sizeof(gpc_point_xyzrgb) * (size_t)n));
        hipError_t e = hipMemcpyAsync(d_cloud, cloud, sizeof(gpc_point_xyzrgb) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(d_cloud);
            return gpc_fail(ctx, GPC_EHIP, "gpc_project_cloud: upload failed: %s", hipGetErrorString(e));
        }
    }
    const int rc = gpc_project_cloud_dev(ctx, (const gpc_point_xyzrgb*)d_cloud, n, res, sz, out);
    if (d_cloud) {
        (void)hipStreamSynchronize(ctx->stream);
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
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const gpc_patches_view& v = p->v;
    const size_t P = (size_t)v.P, N = (size_t)v.n_total;
    struct { void* dst; const void* src; size_t bytes; } cp[10] = {
        {off, v.off, 4 * (P + 1)}, {x0, v.x0, 8 * N}, {x1, v.x1, 8 * N}, {y, v.y, 8 * N}, {rgb, v.rgb, 24 * N},
        {rotations, v.rotations, 72 * P}, {means, v.means, 24 * P}, {rgb_means, v.rgb_means, 24 * P}, {W, v.W, P * (size_t)v.m},
        {src, v.src, 4 * N}};
    for (auto& c : cp)
        if (c.dst && c.bytes) GPC_HIP(ctx, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, ctx->stream));
    GPC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPC_OK;
}

}  // extern "C"
