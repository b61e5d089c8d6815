// dense_mfma_w1.hip -- batched dense GP, n <= 256, depth plane: ONE WAVE PER PATCH, eight patches in flight per CU (gfx950).
//
// Same computation as dense_mfma.hip / dense_mfma_big.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45) and the same tiled LEFT-looking Cholesky as dense_mfma_big.hip (four tile
// columns per step, factor as MFMA operand images in a workspace slot), but a patch is the business of a single wave from its
// first load to its last store:
//
//   * no hand-over, no flag polling, no workgroup barrier, no LDS atomics: the counters of the two-wave shape (round 3) showed
//     its waves waiting 37 % of their time on each other, the kernel at 0.33 of the FP64 peak and NOT bound by its factor stream
//     (every operand load redirected to one L1-resident tile: 3 % faster).  What a single wave cannot overlap inside itself --
//     the dependent pivots of a diagonal factor, a load's latency -- the SIMD's second wave, another patch in another phase, fills.
//   * the step's 4 x 4 diagonal block never leaves the registers: its ten tiles are accumulated in one sweep over j (the four
//     row operands fetched once per j for all ten), factored in place (mf_diag_factor, TRSM, update as register tiles), and the six
//     strictly lower tiles stay in registers as the operand images the row passes multiply by.
//   * the forward solve rides on the sweep (the row operands L_(k+c)j are in registers there: four FMAs per lane give
//     sum_j L_(k+c)j z_j), so the factor is read once per use and nowhere twice in a step except as the column operands of the
//     row passes.
//   * 64-thread workgroups, 15.6 KB of LDS (points, z / alpha, the four L_cc^-1 images), 256 VGPRs: eight workgroups per CU.
//
// Layouts, lane maps and the diagonal factor are those of mfma_tile.h; tiles are kept TRANSPOSED in the C/D layout (see
// dense_mfma.hip), so a TRSM result is directly the operand image of L_ik.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"

#define W1_NPAD 256
#define W1_C 4          // tile columns per step
#define W1_NDT 10       // tiles of a step's diagonal block
#define W1_NT (W1_NPAD / 16)                 // tile rows of a slot
#define W1_TRI (W1_NT * (W1_NT + 1) / 2)     // images of a slot
// offset (doubles) of tile (i, j), j <= i, in a slot
#define W1_TILE(i, j) (((size_t)(i) * (size_t)((i) + 1) / 2 + (size_t)(j)) * MF_IMG)
// Diagnostic build -DW1_EXP_HOT (results wrong by construction): every j-indexed operand load reads tile column 0 -- what the
// latency of the factor stream costs
#ifdef W1_EXP_HOT
__device__ static __forceinline__ int w1_exp_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }   // (not hoistable)
#define W1_JX(j) w1_exp_zero()
#else
#define W1_JX(j) (j)
#endif

// Non-temporal hints on the streams that are used once (-DW1_HINT=mask: 1 the backward solve's loads, 2 the row operands of the passes,
// 4 the stores of the factor tiles), so that the block rows a step reads three times (sweep, column operands of two passes) have a better
// chance to stay in L2.  Measured on one box, C2: none 1.738 ms, backward loads 1.680 (-3.4 %), row operands 1.755 (they ARE read again,
// one step later), stores 1.742, all three 1.727: the backward solve's loads carry the hint.
#ifndef W1_HINT
#define W1_HINT 1
#endif
#define W1_LOAD_BACK(p, l) ((W1_HINT & 1) ? mf_img_load_nt(p, l) : mf_img_load(p, l))
#define W1_LOAD_ROWOP(p, l) ((W1_HINT & 2) ? mf_img_load_nt(p, l) : mf_img_load(p, l))
#define W1_STORE_TILE(p, l, v)                                                                                        \
    do {                                                                                                             \
        if (W1_HINT & 4) mf_img_store_nt(p, l, v);                                                                     \
        else mf_img_store(p, l, v);                                                                                  \
    } while (0)

struct W1Params {
    DenseArgs a;
    double c_exp;
    double pivot_tol;
    double* ws;         // factor slots, one per patch of the launch: W1_TRI images, the lower triangle packed row-major -- tile (i, j) at
                        // (i (i + 1) / 2 + j) * 256 -- which is the layout dense_variance_kernel<16> reads (dense_variance.hip)
    double* linvt;      // the L_kk^-T images: [patch][16][256]
    int export_factor;  // predictive variance: the L_kk^-1 images go to the diagonal positions of the slot
    unsigned long long* stamps;   // diagnostic build (-DW1_STAMPS, GPC_W1_STAMPS=1): [phase] s_memtime sums over all patches
};
// phases: 0 load | 1 sweep: Gram tiles | 2 sweep: j loop | 3 chain | 4 forward solve | 5 pass: first loads + Gram tiles | 6 pass: j loop |
//         7 pass: TRSMs + stores | 8 end-of-step fence | 9 backward | 10 predict | 11 (spare)
#define W1_NPH 12
#ifdef W1_STAMPS
#define W1_STAMP(ph)                                                                                                 \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                                              \
        st_acc_[ph] += t_now_ - t_prev_;                                                                             \
        t_prev_ = t_now_;                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#else
#define W1_STAMP(ph) do { } while (0)
#endif

// One wave per workgroup: LDS instructions of a wave execute in program order, so a value written by one lane is visible to the
// loads of every lane that follow it in the instruction stream -- all that is needed between them is that the COMPILER keeps the
// order.  (__syncthreads() would also wait for every global load and store in flight, i.e. for the prefetches.)
#define W1_LDS_SYNC()                                                                                                \
    do {                                                                                                             \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                                       \
        __builtin_amdgcn_wave_barrier();                                                                             \
    } while (0)

// compile-time loop: f(std::integral_constant<int, I>) for I = A .. B-1 (ring-buffer slots and stream positions must be constants
// for the ring to live in registers)
template <int A, int B, typename F>
__device__ static __forceinline__ void w1_static_for(F&& f)
{
    if constexpr (A < B) {
        f(std::integral_constant<int, A>{});
        w1_static_for<A + 1, B>(f);
    }
}
// stream position -> column index from the end: the largest kk with kk (kk + 1) / 2 <= q
__host__ __device__ constexpr int w1_stream_col(int q)
{
    int kk = 0;
    while ((kk + 1) * (kk + 2) / 2 <= q) ++kk;
    return kk;
}
#define W1_BW 20     // images in flight in the backward solve's ring (160 VGPRs)

__device__ static __forceinline__ d4 w1_mfma4_neg(d4 a, d4 b, d4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 1);   // blgp = 1: NEG(A)
    return acc;
}
__device__ static __forceinline__ d4 w1_trsm(d4 lv, d4 src)
{
    const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
    const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], src[0], z4, 0, 0, 0);
    const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], src[1], z4, 0, 0, 0);
    const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], src[2], z4, 0, 0, 0);
    const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], src[3], z4, 0, 0, 0);
    return (D0 + D1) + (D2 + D3);
}

// Gram tiles (tile row r, tile columns c0 .. c0 + CNT - 1) in the transposed C/D layout: register q of lane l of tile t =
// K[16 r + (l & 15)][16 (c0 + t) + (l >> 4) + 4 q]  (+ the noise diagonal, identity padding beyond n).  The bare values of all CNT
// tiles -- distance, exponential, nothing else -- come first, in ONE basic block: 4 CNT independent dependency chains for the
// scheduler (evaluated tile by tile behind per-tile mode branches the Gram tiles ran at a fifth of the VALU rate: stamps, round 3).
// The noise diagonal (DIAG: the row's last tile is the diagonal tile r == c0 + CNT - 1) and the padding are wave-uniform fix-ups.
template <bool SMALL, int CNT, bool DIAG>
__device__ static __forceinline__ void w1_gram_row(d4 (&v)[W1_C], const double* px0, const double* px1, const double* T, double sf, double cexp,
                                                   double noise, bool dbl, int n, int r, int c0, int lr, int lg)
{
    const int pi = MF_TS * r + lr;
    const double xi0 = px0[pi], xi1 = px1[pi];
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pj = MF_TS * (c0 + t) + lg + 4 * q;
#ifdef W1_EXP_NOGRAM       // diagnostic (results wrong by construction, the matrix stays SPD): what the Gram evaluations cost
            v[t][q] = (pi == pj) ? sf : 1e-3 * sf;
#else
            v[t][q] = SMALL ? gpc_rbf_small(sf, cexp, xi0, xi1, px0[pj], px1[pj]) : gpc_rbf_neg(sf, cexp, xi0, xi1, px0[pj], px1[pj], T);
#endif
        }
    }
    if constexpr (DIAG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // pi == pj on the diagonal tile: lr == lg + 4 q
            double d = v[CNT - 1][q] + noise;            // covariance_matrix(..., training)   gaussian_process.cpp:59-61
            if (dbl) d += noise;                         // C.diagonal() += sigman_sq          :21
            v[CNT - 1][q] = (lr == lg + 4 * q) ? d : v[CNT - 1][q];
        }
    }
    if (MF_TS * (r + 1) > n || MF_TS * (c0 + CNT) > n) {
#pragma unroll
        for (int t = 0; t < CNT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pj = MF_TS * (c0 + t) + lg + 4 * q;
                if (pi >= n || pj >= n) v[t][q] = (pi == pj) ? 1.0 : 0.0;      // identity padding
            }
        }
    }
}

__global__ __launch_bounds__(64, 2) void dense_w1_kernel(W1Params g)
{
    __shared__ __attribute__((aligned(16))) double T[GPC_EXP_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) double px0[W1_NPAD], px1[W1_NPAD], zv[W1_NPAD];     // zv: z, then alpha in place
    __shared__ __attribute__((aligned(16))) double rsbuf[32], wsc[32];
    __shared__ __attribute__((aligned(16))) double LinvC[W1_C * MF_IMG];

    const DenseArgs& A = g.a;
    // The lane id goes through an empty asm at the head of every phase (W1_FRESH_LANE): hipcc hoists lane-dependent address arithmetic
    // out of the surrounding loops -- out of the persistent patch loop too -- and keeps hundreds of such values in registers (398
    // spilled VGPRs in the first version of this kernel); behind an opaque copy they are cheap values recomputed where they are used.
    int lane = threadIdx.x;
    int lr = lane & 15, lg = lane >> 4;
#define W1_FRESH_LANE()                                                                                              \
    do {                                                                                                             \
        lane = mf_opaque(lane);                                                                                      \
        lr = lane & 15;                                                                                              \
        lg = lane >> 4;                                                                                              \
    } while (0)
    const int m = A.m;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    const bool dbl = A.prm.ref_double_noise != 0;
    // ONE patch per workgroup and one factor slot per patch (no persistent patch loop: hipcc hoists every loop-invariant value of
    // the body -- lane masks, addresses, grid constants -- in front of such a loop and then spills them: 350 VGPRs in that form)
    const int patch = blockIdx.x;
    double* Lt = g.ws + (size_t)patch * W1_TRI * MF_IMG;          // tiles (i, j): Lt + W1_TILE(i, j)
    double* LinvTg = g.linvt + (size_t)patch * W1_NT * MF_IMG;    // L_kk^-T images

    gpc_exp_table_init(T);
    {
        const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
        const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
        double* fs = A.f_star + (size_t)patch * m;
        __syncthreads();   // (one wave: a fence -- the exponential table is in LDS)
        if (n <= 0 || n > W1_NPAD) {
            for (int p = lane; p < m; p += 64) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (lane == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            return;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);
#ifdef W1_STAMPS
        unsigned long long st_acc_[W1_NPH] = {};
        unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif

        // ---- points into LDS; extent of the patch around its first point (max-norm) bounds every kernel argument ----
        const double xo0 = A.x0[o], xo1 = A.x1[o];
        double dev = 0.0;
#pragma unroll
        for (int u = 0; u < W1_NPAD / 64; ++u) {
            const int i = lane + 64 * u;
            const bool live = i < n;
            const double q0 = live ? A.x0[o + i] : xo0, q1 = live ? A.x1[o + i] : xo1;
            dev = __builtin_fmax(dev, __builtin_fmax(__builtin_fabs(q0 - xo0), __builtin_fabs(q1 - xo1)));
            px0[i] = live ? q0 : 0.0;
            px1[i] = live ? q1 : 0.0;
            zv[i] = live ? A.y[o + i] : 0.0;        // the right-hand side; the forward solve turns it into z column by column, in place
        }
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) dev = __builtin_fmax(dev, __shfl_xor(dev, o_, 64));
        // Small-argument regime (dense_mfma.hip): |c| d^2 <= 2^-5 for every Gram argument / every separable grid factor -> the
        // degree-7 polynomial instead of the table-driven exponential.  Wave-uniform; false for NaN / inf extents.
        bool small_gram, small_grid;
        {
            const double r = dev;
            const double bq = 0.5 * A.grid_res + __builtin_fmax(__builtin_fabs(xo0), __builtin_fabs(xo1)) + r;
            small_gram = __builtin_amdgcn_readfirstlane((int)(-cexp * (8.0 * r * r) <= GPC_EXP_SMALL_MAX)) != 0;
            small_grid = __builtin_amdgcn_readfirstlane((int)(-cexp * (bq * bq) <= GPC_EXP_SMALL_MAX)) != 0;
        }
        W1_LDS_SYNC();
#ifdef W1_EXP_SMALLONLY      // diagnostic: no table-driven path at all (what the code size costs)
#define W1_GRAM_ROW(v, CNT, DIAG, r, c0) w1_gram_row<true, CNT, DIAG>(v, px0, px1, T, sf, cexp, noise, dbl, n, r, c0, lr, lg)
#else
#define W1_GRAM_ROW(v, CNT, DIAG, r, c0)                                                                              \
    do {                                                                                                             \
        if (small_gram) w1_gram_row<true, CNT, DIAG>(v, px0, px1, T, sf, cexp, noise, dbl, n, r, c0, lr, lg);          \
        else w1_gram_row<false, CNT, DIAG>(v, px0, px1, T, sf, cexp, noise, dbl, n, r, c0, lr, lg);                    \
    } while (0)
#endif

        bool bad = false;
        W1_STAMP(0);
        // ---- tiled left-looking Cholesky, four tile columns (k .. k+3) per step ----
        for (int k = 0; k < nt; k += W1_C) {
            const int nc = min(W1_C, nt - k);                              // tile columns of this step
            const int kl = k - 1;
            W1_FRESH_LANE();
            // ---- the diagonal block: T_(k+i)(k+c) = A - sum_{j<k} L_(k+i)j L_(k+c)j^T, tile d = i (i + 1) / 2 + c, one sweep over j;
            //      the forward-solve sums  part_c = sum_{j<k} L_(k+c)j z_j  from the same operands ----
            const double* rrow[W1_C];
#pragma unroll
            for (int i = 0; i < W1_C; ++i) rrow[i] = Lt + W1_TILE(k + min(i, nc - 1), 0);
            d4 op[2][W1_C], tacc[W1_NDT];
            double part[W1_C];
#pragma unroll
            for (int i = 0; i < W1_C; ++i) {
                op[0][i] = op[1][i] = d4{0.0, 0.0, 0.0, 0.0};
                part[i] = 0.0;
            }
            if (k > 0) {
#pragma unroll
                for (int i = 0; i < W1_C; ++i) op[0][i] = mf_img_load(rrow[i], lane);
            }
#pragma unroll
            for (int d = 0; d < W1_NDT; ++d) tacc[d] = d4{0.0, 0.0, 0.0, 0.0};
            {
                // block row bi: tiles (k + bi, k .. k + bi), the last one the diagonal tile
                d4 gv[W1_C];
                W1_GRAM_ROW(gv, 1, true, k, k);
                tacc[0] = gv[0];
                if (nc > 1) {
                    W1_GRAM_ROW(gv, 2, true, k + 1, k);
                    tacc[1] = gv[0]; tacc[2] = gv[1];
                }
                if (nc > 2) {
                    W1_GRAM_ROW(gv, 3, true, k + 2, k);
                    tacc[3] = gv[0]; tacc[4] = gv[1]; tacc[5] = gv[2];
                }
                if (nc > 3) {
                    W1_GRAM_ROW(gv, 4, true, k + 3, k);
                    tacc[6] = gv[0]; tacc[7] = gv[1]; tacc[8] = gv[2]; tacc[9] = gv[3];
                }
            }
            W1_STAMP(1);
            for (int j = 0; j < k; j += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int jn = min(j + h + 1, kl);
#pragma unroll
                    for (int i = 0; i < W1_C; ++i) op[h ^ 1][i] = mf_img_load(rrow[i] + (size_t)W1_JX(jn) * MF_IMG, lane);
#pragma unroll
                    for (int d = 0; d < W1_NDT; ++d) {
                        const int bi = d >= 6 ? 3 : d >= 3 ? 2 : d >= 1 ? 1 : 0, bc = d - bi * (bi + 1) / 2;
                        if (bi < nc) tacc[d] = w1_mfma4_neg(op[h][bc], op[h][bi], tacc[d]);
                    }
                    const double* zq = zv + MF_TS * (j + h) + lg;
                    const double z0 = zq[0], z1 = zq[4], z2 = zq[8], z3 = zq[12];
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) part[c] += (op[h][c][0] * z0 + op[h][c][1] * z1) + (op[h][c][2] * z2 + op[h][c][3] * z3);
                }
            }
            W1_STAMP(2);
            // ---- the chain, in registers: factor (k,k); then row by row  L_ic = (T_ic - sum_{c2<c} L_ic2 L_cc2^T) L_cc^-T,
            //      T_ii -= sum_c L_ic L_ic^T, factor.  Lb[i (i-1)/2 + c] = operand image of L_(k+i)(k+c), c < i. ----
            W1_FRESH_LANE();
            d4 Lb[W1_C * (W1_C - 1) / 2];
#pragma unroll
            for (int q = 0; q < W1_C * (W1_C - 1) / 2; ++q) Lb[q] = d4{0.0, 0.0, 0.0, 0.0};
            bool ok = mf_diag_factor<true>(tacc[0], rsbuf, LinvC, LinvTg + (size_t)k * MF_IMG, g.pivot_tol, lane);
            W1_LDS_SYNC();
            if (g.export_factor) mf_img_store(Lt + W1_TILE(k, k), lane, mf_img_load(LinvC, lane));
#pragma unroll
            for (int i = 1; i < W1_C; ++i) {
                if (ok && i < nc) {
#pragma unroll
                    for (int c = 0; c < i; ++c) {
                        d4 Tt = tacc[i * (i + 1) / 2 + c];
#pragma unroll
                        for (int c2 = 0; c2 < c; ++c2) Tt = w1_mfma4_neg(Lb[c * (c - 1) / 2 + c2], Lb[i * (i - 1) / 2 + c2], Tt);
                        const d4 lvc = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
                        const d4 L = w1_trsm(lvc, Tt);                     // operand image of L_(k+i)(k+c)
                        Lb[i * (i - 1) / 2 + c] = L;
                        W1_STORE_TILE(Lt + W1_TILE(k + i, k + c), lane, L);
                    }
                    d4 Dii = tacc[i * (i + 1) / 2 + i];
#pragma unroll
                    for (int c = 0; c < i; ++c) Dii = w1_mfma4_neg(Lb[i * (i - 1) / 2 + c], Lb[i * (i - 1) / 2 + c], Dii);
                    ok = mf_diag_factor<true>(Dii, rsbuf, LinvC + i * MF_IMG, LinvTg + (size_t)(k + i) * MF_IMG, g.pivot_tol, lane);
                    W1_LDS_SYNC();
                    if (g.export_factor) mf_img_store(Lt + W1_TILE(k + i, k + i), lane, mf_img_load(LinvC + i * MF_IMG, mf_opaque(lane)));
                }
            }
#if defined(W1_EXP_HOT) || defined(W1_EXP_NOPASSTRSM) || defined(W1_EXP_NOBACK)
            ok = true;
#endif
            if (!ok) { bad = true; break; }
            W1_STAMP(3);
            W1_FRESH_LANE();
            // ---- forward solve of the step's columns: z_(k+c) = L_cc^-1 (y_(k+c) - part_c - sum_{c2<c} L_(k+c)(k+c2) z_(k+c2)) ----
            // 16 x 16 mat-vecs on the VALU from the operand images: lane l holds M[l & 15][(l >> 4) + 4 s] -- the lane's partial sum over
            // its four columns, then the four lane groups of a row are added.
#pragma unroll
            for (int c = 0; c < W1_C; ++c) {
                if (c < nc) {
                    const int pj = MF_TS * (k + c) + lr;
                    double pp = part[c];
#pragma unroll
                    for (int c2 = 0; c2 < c; ++c2) {
                        const d4 lb = Lb[c * (c - 1) / 2 + c2];
                        const double* zq = zv + MF_TS * (k + c2) + lg;
                        pp += (lb[0] * zq[0] + lb[1] * zq[4]) + (lb[2] * zq[8] + lb[3] * zq[12]);
                    }
                    pp += __shfl_xor(pp, 16, 64);
                    pp += __shfl_xor(pp, 32, 64);
                    const double yv = zv[pj];                 // y (0 beyond n), loaded with the points
                    W1_LDS_SYNC();
                    if (lg == 0) zv[pj] = yv - pp;            // t_c, where z_c goes next
                    W1_LDS_SYNC();
                    const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
                    const double* tq = zv + MF_TS * (k + c) + lg;
                    double zz = (lv[0] * tq[0] + lv[1] * tq[4]) + (lv[2] * tq[8] + lv[3] * tq[12]);
                    zz += __shfl_xor(zz, 16, 64);
                    zz += __shfl_xor(zz, 32, 64);
                    W1_LDS_SYNC();
                    if (lg == 0) zv[pj] = zz;
                    W1_LDS_SYNC();
                }
            }
            W1_STAMP(4);
            const int rows_tot_ = nt - (k + nc);
            if (rows_tot_ > 0 && !(k == 0 && nc == W1_C)) __syncthreads();     // the chain's tiles are in the workspace before the row passes read them back
            // ---- rows k + nc .. nt - 1, two per pass: update the four accumulators over j < k, then column by column
            //      T_r(k+c) -= sum_{c2<c} L_r(k+c2) L_(k+c)(k+c2)^T,  L_r(k+c) = T L_cc^-T ----
            const int rows_tot = rows_tot_;
            int first_row0 = 0;
            if (k == 0 && nc == W1_C) {
                // Step 0 has no update loop, so its registers are free: FOUR rows per pass with the block's lower tiles resident (no
                // reload from the workspace -- at eight patches per CU those reloads miss L2: 36 of a patch's 512 image reads) and
                // four independent TRSM chains interleaved.
                for (; first_row0 < rows_tot; first_row0 += 4) {
                    const int np4 = min(4, rows_tot - first_row0);
                    W1_FRESH_LANE();
                    d4 a4[W1_C][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r_ = W1_C + first_row0 + min(t, np4 - 1);       // (t >= np4: a copy of the last row, never stored)
                        d4 gv[W1_C];
                        W1_GRAM_ROW(gv, 2, false, r_, 0);
                        a4[0][t] = gv[0]; a4[1][t] = gv[1];
                        __builtin_amdgcn_sched_barrier(0);
                        W1_GRAM_ROW(gv, 2, false, r_, 2);
                        a4[2][t] = gv[0]; a4[3][t] = gv[1];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    W1_STAMP(5);
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) {
                        const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
#pragma unroll
                            for (int c2 = 0; c2 < c; ++c2) a4[c][t] = w1_mfma4_neg(Lb[c * (c - 1) / 2 + c2], a4[c2][t], a4[c][t]);
                            a4[c][t] = w1_trsm(lv, a4[c][t]);
                            if (t < np4) W1_STORE_TILE(Lt + W1_TILE(W1_C + first_row0 + t, c), lane, a4[c][t]);
                        }
                    }
                    W1_STAMP(7);
                }
            }
            // k > 0 (rows beyond the block exist only when the block has all four columns): FOUR rows per pass -- 16 accumulators; the four
            // column operands L_(k+c)j double-buffered, the four row operands L_rj single-buffered and re-requested for j + 1 as soon as
            // their four products of j are issued (twelve products of lead).  Eight images per 16 products: the two-row passes of the
            // first version fetched six per eight, and at eight patches per CU every one of them is an L2 miss (the kernel moves
            // 10.7 GB per launch at 6 TB/s: it runs at the HBM bandwidth the part delivers).
            for (int first_row = first_row0; first_row < rows_tot; first_row += 4) {
                const int np4 = min(4, rows_tot - first_row);
                W1_FRESH_LANE();
                int rr[4];
                const double* rw_[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    rr[t] = k + W1_C + first_row + min(t, np4 - 1);        // (t >= np4: a copy of the last row, never stored)
                    rw_[t] = Lt + W1_TILE(rr[t], 0);
                }
                d4 acc[W1_C][4], A2[2][W1_C], B[4];
#pragma unroll
                for (int c = 0; c < W1_C; ++c) A2[0][c] = mf_img_load(rrow[c], lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) B[t] = W1_LOAD_ROWOP(rw_[t], lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    d4 gv[W1_C];
                    W1_GRAM_ROW(gv, 2, false, rr[t], k);
                    acc[0][t] = gv[0]; acc[1][t] = gv[1];
                    __builtin_amdgcn_sched_barrier(0);
                    W1_GRAM_ROW(gv, 2, false, rr[t], k + 2);
                    acc[2][t] = gv[0]; acc[3][t] = gv[1];
                    __builtin_amdgcn_sched_barrier(0);
                }
                W1_STAMP(5);
                d4 Lq[W1_C * (W1_C - 1) / 2];      // images of L_(k+i)(k+c2), c2 < i, at i (i - 1) / 2 + c2
                // one j: request the column operands of jn into the other stage, then row by row the four products and the row's next request
#define W1_PASS_J(st, jn, PREFETCH)                                                                                  \
    do {                                                                                                             \
        /* (the scheduling barriers pin the requests where they are written: left alone, hipcc sinks every one of them to its  \
           first use -- request, s_waitcnt vmcnt(0), MFMA -- and the prefetch is gone) */                              \
        if (PREFETCH) {                                                                                              \
            _Pragma("unroll") for (int c = 0; c < W1_C; ++c) A2[(st) ^ 1][c] = mf_img_load(rrow[c] + (size_t)W1_JX(jn) * MF_IMG, lane); \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                              \
            _Pragma("unroll") for (int c = 0; c < W1_C; ++c) acc[c][t] = w1_mfma4_neg(A2[st][c], B[t], acc[c][t]);    \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (PREFETCH) B[t] = W1_LOAD_ROWOP(rw_[t] + (size_t)W1_JX(jn) * MF_IMG, lane);                           \
            else {     /* the last j: the block's lower tiles for the TRSMs take the place of the operands that are done */ \
                if (t < 3) Lq[t] = mf_img_load(Lt + W1_TILE(k + (t == 0 ? 1 : 2), k + (t == 2 ? 1 : 0)), lane);      \
                else {                                                                                               \
                    _Pragma("unroll") for (int c2 = 0; c2 < 3; ++c2) Lq[3 + c2] = mf_img_load(Lt + W1_TILE(k + 3, k + c2), lane); \
                }                                                                                                    \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    } while (0)
                for (int j = 0; j + 2 < k; j += 2) {
                    W1_PASS_J(0, j + 1, true);
                    W1_PASS_J(1, j + 2, true);
                }
                W1_PASS_J(0, kl, true);                 // (k is a multiple of four: the last pair of j; its second half requests nothing)
                W1_PASS_J(1, kl, false);
#undef W1_PASS_J
                W1_STAMP(6);
                W1_FRESH_LANE();
                // (the block's strictly lower tiles Lq came back from the workspace during the last j -- this wave's own stores of the
                // chain, fenced below it: as registers they would be live across the update loop, 48 VGPRs on top of its 224)
#pragma unroll
                for (int c = 0; c < W1_C; ++c) {
                    const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#ifndef W1_EXP_NOPASSTRSM      // (diagnostic: what the TRSM chains of the row passes cost -- stores only)
#pragma unroll
                        for (int c2 = 0; c2 < c; ++c2) acc[c][t] = w1_mfma4_neg(Lq[c * (c - 1) / 2 + c2], acc[c2][t], acc[c][t]);
                        acc[c][t] = w1_trsm(lv, acc[c][t]);
#endif
                        if (t < np4) W1_STORE_TILE(Lt + W1_TILE(rr[t], k + c), lane, acc[c][t]);
                    }
                }
                W1_STAMP(7);
            }
            __syncthreads();   // the column block is in the workspace (this wave's own stores, read back by its next sweep)
            W1_STAMP(8);
        }

        if (bad) {
            for (int p = lane; p < m; p += 64) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = lane; i < n; i += 64) A.alpha_out[o + i] = __builtin_nan("");
            if (lane == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            return;
        }

#ifndef W1_EXP_NOBACK
        // ---- backward solve L^T alpha = z, tile columns from the last to the first; alpha replaces z in place ----
        // Column k: w_k = sum_{i>k} L_ik^T alpha_i on the VALU (the products contract over the ROW index, which the image layout
        // cannot feed to an MFMA) with the transposing DPP row reduction, then alpha_k = L_kk^-T (z_k - w_k): four MFMAs.
        // The tiles do not depend on alpha, and the factor of the 2048 patches in flight (620 MB) is in HBM: the solve is ONE STREAM
        // of nt (nt + 1) / 2 images -- column by column, each column's tiles and then its L_kk^-T -- consumed in order through a ring
        // of W1_BW registers images that keeps W1_BW requests in flight (with one column of look-ahead this phase took 12 % of the
        // kernel for 2 % of its arithmetic: every column waited out an HBM round trip).  Everything is unrolled over the stream
        // positions of a 16-column factor, aligned at its last column: column kk from the end (k = nt-1-kk) has kk tiles, rows
        // k+1+t; positions of columns a smaller factor does not have are skipped, their requests clamped to the slot's first image.
        {
            W1_FRESH_LANE();
            constexpr int NP = W1_NPAD / MF_TS;                     // 16
            constexpr int SLEN = NP * (NP + 1) / 2;                 // 136 stream positions
            d4 win[W1_BW];
            auto stream_addr = [&](auto P) __attribute__((always_inline)) -> const double* {
                constexpr int q = decltype(P)::value;
                constexpr int kk = w1_stream_col(q), t = q - kk * (kk + 1) / 2;
                const int kq = max(nt - 1 - kk, 0);                  // (clamped for a column this factor does not have)
                const double* ad = (t < kk) ? Lt + W1_TILE(kq + 1 + t, kq) : LinvTg + (size_t)kq * MF_IMG;
                return kk < nt ? ad : Lt;
            };
            w1_static_for<0, W1_BW>([&](auto P) __attribute__((always_inline)) {
                constexpr int q = decltype(P)::value;
                win[q % W1_BW] = W1_LOAD_BACK(stream_addr(P), lane);
            });
            d4 pa = d4{0.0, 0.0, 0.0, 0.0};
            w1_static_for<0, SLEN>([&](auto P) __attribute__((always_inline)) {
                constexpr int q = decltype(P)::value;
                constexpr int kk = w1_stream_col(q), t = q - kk * (kk + 1) / 2;
                if (kk < nt) {
                    const int k = nt - 1 - kk;
                    if constexpr (t < kk) {
                        const double a_ = zv[MF_TS * (k + 1 + t) + lr];
                        pa += win[q % W1_BW] * a_;                       // the image of L_ik: [l & 15][(l >> 4) + 4 s]
                    } else {
                        d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                        if constexpr (kk > 0) {
                            const double tot = mf_row_reduce4(pa, lr);   // lanes lr = 0, 4, 8, 12 hold components 0 .. 3
                            if ((lr & 3) == 0) wsc[lg + 4 * (lr >> 2)] = tot;
                            W1_LDS_SYNC();
                            if (lr == 0) {
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) ub[q4] = zv[MF_TS * k + lg + 4 * q4] - wsc[lg + 4 * q4];
                            }
                        } else {
                            if (lr == 0) {
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) ub[q4] = zv[MF_TS * k + lg + 4 * q4];
                            }
                        }
                        const d4 al = w1_trsm(win[q % W1_BW], ub);       // lanes lr = 0: alpha[16 k + (l >> 4) + 4 r]
                        W1_LDS_SYNC();
                        if (lr == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) zv[MF_TS * k + lg + 4 * r] = al[r];
                        }
                        W1_LDS_SYNC();
                        pa = d4{0.0, 0.0, 0.0, 0.0};
                    }
                }
                if constexpr (q + W1_BW < SLEN) {
                    win[q % W1_BW] = W1_LOAD_BACK(stream_addr(std::integral_constant<int, q + W1_BW>{}), lane);
                }
            });
        }
#endif
        W1_STAMP(9);
        double* av = zv;
        if (A.alpha_out)
            for (int i = lane; i < n; i += 64) A.alpha_out[o + i] = av[i];

#ifndef W1_EXP_NOPRED
        // ---- predictive mean ----
        W1_FRESH_LANE();
        if (m <= 0) {
            // (the variance solve forms the mean from its own K* tiles: the fit predicts nothing)
        } else if (A.xs0 == nullptr && A.grid_sz <= 32) {
            // separable grid: f[py][px] = sum_i Ey[py][i] * (sf alpha_i Ex[px][i]): four 16 x 16 output tiles, no reduction across waves
            const int sz = A.grid_sz;
            const double res = A.grid_res;
            d4 P[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) P[mt][nl] = d4{0.0, 0.0, 0.0, 0.0};
            double gq[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) gq[h] = res * (((double)(16 * h + lr) + 0.5) / (double)sz - 0.5);
            // No masks in the loop: a point beyond n contributes nothing because its weight sf alpha_i is SELECTED to zero (alpha is
            // zero from n to 16 nt -- identity padding solves to 0 -- but never written beyond: select, do not multiply), and the
            // grid rows / columns beyond sz are finite numbers that are never stored.  One basic block per 32 points: 32 independent
            // exponentials for the scheduler (behind per-element masks and mode branches this phase took a sixth of the kernel).
            auto predict_loop = [&](auto small_tag) __attribute__((always_inline)) {
                constexpr bool SM = decltype(small_tag)::value;
                for (int ibase = 0; ibase < n; ibase += 32) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ibase + 4 * s + lg;
                        const int ic = min(i, W1_NPAD - 1);
                        const double al = (i < n) ? sf * av[ic] : 0.0;
                        const double xi0 = px0[ic], xi1 = px1[ic];
                        double ea[2], eb[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double dy = gq[h] - xi1, dx = gq[h] - xi0;
                            ea[h] = SM ? gpc_exp_small(cexp * (dy * dy)) : gpc_exp_neg(cexp * (dy * dy), T);            // Ey[py][i]
                            eb[h] = (SM ? gpc_exp_small(cexp * (dx * dx)) : gpc_exp_neg(cexp * (dx * dx), T)) * al;     // Ex[px][i] * sf alpha_i
                        }
#pragma unroll
                        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                P[mt][nl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[mt], eb[nl], P[mt][nl], 0, 0, 0);
                    }
                }
            };
            if (small_grid) predict_loop(std::true_type{});
            else predict_loop(std::false_type{});
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int py = 16 * mt + lg + 4 * r, pxx = 16 * nl + lr;
                        if (py < sz && pxx < sz) fs[py * sz + pxx] = P[mt][nl][r];
                    }
        } else {
            // point-wise X* (or a grid wider than 32): one lane per prediction point
            for (int p = lane; p < m; p += 64) {
                double q0, q1;
                if (A.xs0) {
                    q0 = A.xs0[p];
                    q1 = A.xs1[p];
                } else {
                    const int gx = p % A.grid_sz, gy = p / A.grid_sz;
                    q0 = A.grid_res * (((double)gx + 0.5) / (double)A.grid_sz - 0.5);
                    q1 = A.grid_res * (((double)gy + 0.5) / (double)A.grid_sz - 0.5);
                }
                double s_ = 0.0;
                for (int i = 0; i < n; ++i) s_ += gpc_rbf_neg(sf, cexp, px0[i], px1[i], q0, q1, T) * av[i];
                fs[p] = s_;
            }
        }
#endif
        if (lane == 0 && A.status) A.status[patch] = GPC_STATUS_OK;
#ifdef W1_STAMPS
        W1_STAMP(10);
        if (g.stamps && lane == 0)
            for (int q_ = 0; q_ < W1_NPH; ++q_) atomicAdd(g.stamps + q_, st_acc_[q_]);
#endif
    }
}

bool dense_w1_supported(const DenseArgs& a)
{
    // (with the variance: point-wise X* only -- the variance entry has no grid form)
    return a.n_max <= W1_NPAD && a.ny == 1 && !a.sel && (a.v_star == nullptr || a.xs0 != nullptr);
}

// One factor slot per patch of a launch (304 KB: 2.5 GB for the 8192 patches of BASELINE config 2 -- sized for 288 GB); a larger
// batch goes through in launches of W1_MAX_SLOTS patches that reuse the slots (a launch of 8192 patches is four rounds of the 2048
// resident workgroups: its ramp and tail are ~3 % of it).  `cap` > 0: the dispatcher's retry with fewer slots after GPC_ENOMEM.
#define W1_MAX_SLOTS 8192
static int w1_chunk(const DenseArgs& a, int cap_in)
{
    const char* e = getenv("GPC_W1_SLOTS");
    int cap = e && atoi(e) > 0 ? atoi(e) : W1_MAX_SLOTS;
    if (cap_in > 0 && cap_in < cap) cap = cap_in;
    return a.P < cap ? a.P : cap;
}

size_t dense_w1_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out, int cap)
{
    (void)ctx;
    const int grid = w1_chunk(a, cap);
    if (grid_out) *grid_out = grid;
    // factor slots | L_kk^-T images | (variance without alpha_out: the weights the variance kernel forms the mean from)
    return sizeof(double) * ((size_t)(W1_TRI + W1_NT) * MF_IMG * (size_t)grid + (a.v_star ? (size_t)a.n_total : 0));
}

int dense_w1_launch(gpc_ctx* ctx, const DenseArgs& a_in, int grid)
{
    DenseArgs a = a_in;
    double* v_star = a.v_star;
    a.v_star = nullptr;
    W1Params g;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.ws = static_cast<double*>(ctx->ws);
    g.linvt = g.ws + (size_t)W1_TRI * MF_IMG * (size_t)grid;
    g.export_factor = v_star ? 1 : 0;
    g.stamps = nullptr;
    if (v_star) {
        // Predictive variance (gaussian_process::predict_measurements, /root/reference/src/gaussian_process.cpp:35-43): the slots are the
        // factor export dense_variance_kernel<16> reads, the fit predicts nothing (the solve forms the mean from the same K* tiles)
        a.m = 0;
        if (!a.alpha_out) a.alpha_out = g.linvt + (size_t)W1_NT * MF_IMG * (size_t)grid;
    }
    ctx->last_dense_kernel = v_star ? "dense_mfma_w1 + dense_variance" : "dense_mfma_w1";
#ifdef W1_STAMPS
    if (getenv("GPC_W1_STAMPS")) {
        GPC_HIP(ctx, hipMalloc(&g.stamps, sizeof(unsigned long long) * W1_NPH));
        GPC_HIP(ctx, hipMemsetAsync(g.stamps, 0, sizeof(unsigned long long) * W1_NPH, ctx->stream));
    }
#endif
    // (diagnostic: GPC_W1_LDS_PAD bytes of unused dynamic LDS per workgroup cap the workgroups resident on a CU)
    const char* pad_e = getenv("GPC_W1_LDS_PAD");
    const size_t pad = pad_e ? (size_t)atoi(pad_e) : 0;
    for (int base = 0; base < a.P; base += grid) {
        // a launch works on patches base .. base + cnt - 1: the kernel's patch index is its workgroup index; `off`, f*, V* and status
        // are passed shifted (off[] holds absolute point offsets, so x, y and alpha stay as they are)
        const int cnt = a.P - base < grid ? a.P - base : grid;
        g.a = a;
        g.a.P = cnt;
        g.a.off = a.off + base;
        g.a.f_star = a.f_star ? a.f_star + (size_t)base * a.ny * a_in.m : nullptr;
        g.a.status = a.status ? a.status + base : nullptr;
        hipLaunchKernelGGL(dense_w1_kernel, dim3(cnt), dim3(64), pad, ctx->stream, g);
        GPC_HIP(ctx, hipGetLastError());
        if (v_star) {
            DenseArgs av = g.a;
            av.m = a_in.m;
            const int rc = dense_variance_launch(ctx, av, W1_NT, g.ws, a.alpha_out, v_star + (size_t)base * a_in.m);
            if (rc != GPC_OK) return rc;
        }
    }
#ifdef W1_STAMPS
    if (g.stamps) {
        unsigned long long h[W1_NPH];
        GPC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GPC_HIP(ctx, hipMemcpy(h, g.stamps, sizeof(h), hipMemcpyDeviceToHost));
        (void)hipFree(g.stamps);
        static const char* names[W1_NPH] = {"load", "sweep gram", "sweep loop", "chain", "forward", "pass gram", "pass loop", "pass trsm",
                                            "step fence", "backward", "predict", ""};
        unsigned long long tot = 0;
        for (int q = 0; q < W1_NPH; ++q) tot += h[q];
        fprintf(stderr, "[GPC_W1_STAMPS] mean s_memtime ticks per patch (total %.0f)\n", (double)tot / a.P);
        for (int q = 0; q < 11; ++q) fprintf(stderr, "  %-12s %9.0f  %5.1f %%\n", names[q], (double)h[q] / a.P, 100.0 * h[q] / tot);
    }
#endif
    return GPC_OK;
}
