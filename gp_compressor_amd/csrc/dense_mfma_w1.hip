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

#define W1_NPAD_MAX 512   // the kernel is instantiated for 256 (eight patches per CU) and 512 points (seven: 21 KB of LDS each)
#define W1_C 4          // tile columns per step
#define W1_NDT 10       // tiles of a step's diagonal block
#define W1_NT_OF(npad) ((npad) / 16)                               // tile rows of a slot
#define W1_TRI_OF(npad) (W1_NT_OF(npad) * (W1_NT_OF(npad) + 1) / 2)  // images of a slot
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
// Round 4 (each with its own switch, so that one library build can be timed against another on the same box):
//   W1_TRSM_CHAIN  the four products of a TRSM chained through the accumulator instead of four independent products + an add tree
//                  (12 VALU adds per TRSM, 136 TRSMs per patch)
//   W1_CMASK       the diagonal factor with constant lane masks in SGPR pairs (mf_diag_factor_c: 4 instead of ~12 VALU per pivot)
//   W1_CARRY       the row pass of a step that covers the NEXT block's rows runs last and its sixteen result tiles stay in registers:
//                  the next step's block takes the products over those four tile columns from them, its sweep reads only the columns
//                  before (48 of the 96 sweep images of a 16-row factor are not read at all)
//   W1_LASTRES     the last step's block (six tiles + four L_cc^-T) goes into the backward solve from registers / LDS and is never
//                  written to the workspace (n a multiple of 64 points' worth of tiles, no factor export)
#ifndef W1_TRSM_CHAIN
#define W1_TRSM_CHAIN 1
#endif
#ifndef W1_CMASK
#define W1_CMASK 1
#endif
// (W1_CARRY and W1_LASTRES are OFF in the shipped build: hipcc answers both with register spills that move more bytes than they save --
//  46 / 386 / 208 spilled VGPRs without / with the carry / with the resident last block, scratch stores and reloads in every step)
#ifndef W1_CARRY
#define W1_CARRY 0
#endif
#ifndef W1_LASTRES
#define W1_LASTRES 0
#endif
#if W1_LASTRES && !W1_CMASK
#error "W1_LASTRES needs the diagonal factor that can skip its L^-T store (W1_CMASK)"
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
    double pivot_tol;   // (already in the scale the factor runs in: see noise_u)
    // host-computed wave-uniform constants (kernel arguments live in SGPRs; computed in the kernel they would be VGPR pairs for its whole
    // length): sqrt(-c), the noise term and the weight multipliers of the scale the factor runs in -- unit scale (K / sigma_f^2: noise_u =
    // noise / sigma_f^2, prediction weights as they are, alpha_out = a' / sigma_f^2) unless the factor is exported for the variance
    double cs, noise_u, sfp, a_out;
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
#if W1_TRSM_CHAIN
    d4 D = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], src[0], z4, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], src[1], D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], src[2], D, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], src[3], D, 0, 0, 0);
#endif
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
// Round 4: MODE 0 = table-driven exponential of c d^2 on the caller's coordinates; MODE 1 / 2 = the patch proved c d^2 >= -2^-5 / -2^-8
// for every pair and holds its coordinates pre-scaled by sqrt(-c): the squared distance IS the argument, exp(-t) is a degree-7 / degree-5
// polynomial with literal coefficients -- 11 / 9 VALU operations per value (round 3: 13).  The kernel is evaluated WITHOUT sigma_f^2
// (unit scale: the factor of K / sigma_f^2, see the kernel body) unless the factor is exported (scale_sf).
template <int MODE, int CNT, bool DIAG>
__device__ static __forceinline__ void w1_gram_row(d4 (&v)[W1_C], const double* px0, const double* px1, const double* T, double sf, bool scale_sf,
                                                   double cexp, double noise, bool dbl, int n, int r, int c0, int lr, int lg)
{
    const int pi = MF_TS * r + lr;
    const double xi0 = px0[pi], xi1 = px1[pi];
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pj = MF_TS * (c0 + t) + lg + 4 * q;
#ifdef W1_EXP_NOGRAM       // diagnostic (results wrong by construction, the matrix stays SPD): what the Gram evaluations cost
            v[t][q] = (pi == pj) ? 1.0 : 1e-3;
#else
            const double d0 = xi0 - px0[pj], d1 = xi1 - px1[pj];
            const double sq = __builtin_fma(d0, d0, d1 * d1);
            v[t][q] = MODE == 0 ? gpc_exp_neg(cexp * sq, T) : gpc_expm_poly<MODE == 1 ? 7 : 5>(sq);
#endif
        }
    }
    if (scale_sf) {
#pragma unroll
        for (int t = 0; t < CNT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] *= sf;
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

template <int W1_NPAD>
__global__ __launch_bounds__(64, 2) void dense_w1_kernel(W1Params g)
{
    constexpr int W1_NT = W1_NT_OF(W1_NPAD), W1_TRI = W1_TRI_OF(W1_NPAD);
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
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp;
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

        // ---- points into LDS; the bounding box of the patch bounds every kernel argument ----
        // Regimes of the exponential (wave-uniform; false for NaN / inf extents):
        //   mode_g (Gram):  2 when |c| d^2 <= 2^-8 for every pair of the patch (box diagonal), 1 when <= 2^-5, else 0
        //   mode_p (separable grid factors): the same thresholds on |c| (grid-to-point distance along one axis)^2; 0 when mode_g is 0
        // mode_g > 0: the coordinates go to LDS centred on the first point and SCALED by sqrt(-c), so that a squared distance is the
        // argument itself (gpc_expm_poly: no multiplication by c, literal coefficients); the rounding of the scaled coordinates moves an
        // argument by <= 2^-50 |c| extent^2 <= 2^-55 here.  mode_g == 0: raw coordinates, table-driven exponential, as in round 3.
        // That is the regime of a GP patch model whose length scale exceeds the patch -- the reference's dense defaults and C2.
        const double xo0 = A.x0[o], xo1 = A.x1[o];
        double qr0[W1_NPAD / 64], qr1[W1_NPAD / 64];
        double mn0 = xo0, mx0 = xo0, mn1 = xo1, mx1 = xo1;
#pragma unroll
        for (int u = 0; u < W1_NPAD / 64; ++u) {
            const int i = lane + 64 * u;
            const bool live = i < n;
            qr0[u] = live ? A.x0[o + i] : xo0;
            qr1[u] = live ? A.x1[o + i] : xo1;
            mn0 = __builtin_fmin(mn0, qr0[u]); mx0 = __builtin_fmax(mx0, qr0[u]);
            mn1 = __builtin_fmin(mn1, qr1[u]); mx1 = __builtin_fmax(mx1, qr1[u]);
            zv[i] = live ? A.y[o + i] : 0.0;        // the right-hand side; the forward solve turns it into z column by column, in place
        }
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) {
            mn0 = __builtin_fmin(mn0, __shfl_xor(mn0, o_, 64)); mx0 = __builtin_fmax(mx0, __shfl_xor(mx0, o_, 64));
            mn1 = __builtin_fmin(mn1, __shfl_xor(mn1, o_, 64)); mx1 = __builtin_fmax(mx1, __shfl_xor(mx1, o_, 64));
        }
        int mode_g, mode_p;
        {
            const bool nan_box = !((mx0 - mn0) + (mx1 - mn1) < __builtin_inf());         // a NaN coordinate slips through fmin / fmax
            const double rx = mx0 - mn0, ry = mx1 - mn1;
            const double tg = nan_box ? __builtin_inf() : -cexp * __builtin_fma(rx, rx, ry * ry);
            const double hb = 0.5 * A.grid_res;
            const double bx = hb + __builtin_fmax(__builtin_fabs(mn0), __builtin_fabs(mx0));
            const double by = hb + __builtin_fmax(__builtin_fabs(mn1), __builtin_fabs(mx1));
            const double bq = __builtin_fmax(bx, by);
            const double tp = nan_box ? __builtin_inf() : -cexp * (bq * bq);
            mode_g = __builtin_amdgcn_readfirstlane(tg <= GPC_EXP_TINY_MAX ? 2 : tg <= GPC_EXP_SMALL_MAX ? 1 : 0);
            mode_p = __builtin_amdgcn_readfirstlane(tp <= GPC_EXP_TINY_MAX ? 2 : tp <= GPC_EXP_SMALL_MAX ? 1 : 0);
            if (mode_g == 0) mode_p = 0;
#ifdef W1_EXP_FORCE_MODE      // diagnostic: the regime chosen by hand (0 table, 1 degree 7, 2 degree 5)
            mode_g = mode_g > 0 ? W1_EXP_FORCE_MODE : 0;
            mode_p = mode_p > 0 ? W1_EXP_FORCE_MODE : 0;
#endif
        }
        const bool scaled = mode_g > 0;
        {
            const double cs = scaled ? g.cs : 1.0;                             // coordinate scale
            const double co0 = scaled ? xo0 : 0.0, co1 = scaled ? xo1 : 0.0;   // ... and centre
#pragma unroll
            for (int u = 0; u < W1_NPAD / 64; ++u) {
                const int i = lane + 64 * u;
                const bool live = i < n;
                px0[i] = live ? cs * (qr0[u] - co0) : 0.0;
                px1[i] = live ? cs * (qr1[u] - co1) : 0.0;
            }
        }
        // Unit scale: K + 2 noise I = sigma_f^2 (E + 2 (noise / sigma_f^2) I) with E the kernel without sigma_f^2.  The wave factors
        // E' = E + ..., solves E' a' = y, and predicts f* = E*^T a' -- sigma_f^2 cancels in the predictive mean, so no kernel value is
        // ever multiplied by it (one VALU operation per value); alpha_out = a' / sigma_f^2.  With the factor exported for the
        // predictive variance the true K is factored (scale_sf): the variance kernel reads L and L_kk^-1 of K itself.
        const bool scale_sf = g.export_factor != 0;
        const double noise_u = g.noise_u, ptol = g.pivot_tol;
        W1_LDS_SYNC();
#define W1_GRAM_ROW(v, CNT, DIAG, r, c0)                                                                              \
    do {                                                                                                             \
        if (mode_g == 2) w1_gram_row<2, CNT, DIAG>(v, px0, px1, T, sf, scale_sf, cexp, noise_u, dbl, n, r, c0, lr, lg);   \
        else if (mode_g == 1) w1_gram_row<1, CNT, DIAG>(v, px0, px1, T, sf, scale_sf, cexp, noise_u, dbl, n, r, c0, lr, lg); \
        else w1_gram_row<0, CNT, DIAG>(v, px0, px1, T, sf, scale_sf, cexp, noise_u, dbl, n, r, c0, lr, lg);            \
    } while (0)

        // One position of the backward solve's stream (see the solve below): column kk from the end (k = nt - 1 - kk) has kk tiles, rows
        // k + 1 + t, then its L_kk^-T.  A tile adds its part of w_k = sum_{i>k} L_ik^T alpha_i on the VALU (the products contract over the
        // ROW index, which the image layout cannot feed to an MFMA); L_kk^-T closes the column: transposing DPP row reduction of w_k,
        // alpha_k = L_kk^-T (z_k - w_k) as four MFMAs, alpha_k into zv in place of z_k.
        auto bw_step = [&](auto P, const d4 img, d4& pa) __attribute__((always_inline)) {
            constexpr int q = decltype(P)::value;
            constexpr int kk = w1_stream_col(q), t = q - kk * (kk + 1) / 2;
            if (kk < nt) {
                const int k = nt - 1 - kk;
                if constexpr (t < kk) {
                    const double a_ = zv[MF_TS * (k + 1 + t) + lr];
                    pa += img * a_;                                      // the image of L_ik: [l & 15][(l >> 4) + 4 s]
                } else {
                    d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                    if constexpr (kk > 0) {
                        const double tot = mf_row_reduce4(pa, lr);       // lanes lr = 0, 4, 8, 12 hold components 0 .. 3
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
                    const d4 al = w1_trsm(img, ub);                      // lanes lr = 0: alpha[16 k + (l >> 4) + 4 r]
                    W1_LDS_SYNC();
                    if (lr == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) zv[MF_TS * k + lg + 4 * r] = al[r];
                    }
                    W1_LDS_SYNC();
                    pa = d4{0.0, 0.0, 0.0, 0.0};
                }
            }
        };
        constexpr int NLAST = W1_C * (W1_C + 1) / 2;   // stream positions of the last block: its six tiles and four L_cc^-T
        bool bad = false;
        W1_STAMP(0);
        // ---- tiled left-looking Cholesky, four tile columns (k .. k+3) per step ----
        // Rotated loop (W1_CARRY): the accumulators of a step's diagonal block, tacc, and its forward-solve sums, part, are set up at
        // the END of the previous step, right behind the row pass of the block's own rows -- that pass runs last, its sixteen result
        // tiles cy[c][t] (the operand images of L_(k+4+t)(k+c)) are still in registers, and the block takes its products over those
        // four tile columns from them.  The step then sweeps only the tile columns before (jend), from the workspace.  What crosses
        // the loop's back edge is tacc and part (88 registers that the sweep needs anyway), never the sixteen tiles.
#if W1_CARRY
        d4 tacc[W1_NDT];
        double part[W1_C];
#endif
        // block row bi of the step at kb (nb tile columns): Gram tiles (kb + bi, kb .. kb + bi), the last one the diagonal tile; WITH_CY:
        // minus the products over the previous step's tile columns, and the forward-solve sums over them (z of those columns is final).
        // From the last row down, so that the carry rows above bi are dead once it is done.
#define W1_BLOCK_ROW(bi, kb, nb, WITH_CY)                                                                            \
    do {                                                                                                             \
        if ((bi) < (nb)) {                                                                                           \
            d4 gv[W1_C];                                                                                             \
            W1_GRAM_ROW(gv, (bi) + 1, true, (kb) + (bi), (kb));                                                      \
            _Pragma("unroll") for (int bc = 0; bc <= (bi); ++bc) tacc[(bi) * ((bi) + 1) / 2 + bc] = gv[bc];           \
            if (WITH_CY) {                                                                                           \
                _Pragma("unroll") for (int c = 0; c < W1_C; ++c) {                                                   \
                    _Pragma("unroll") for (int bc = 0; bc <= (bi); ++bc)                                             \
                        tacc[(bi) * ((bi) + 1) / 2 + bc] = w1_mfma4_neg(cy[c][bc], cy[c][bi], tacc[(bi) * ((bi) + 1) / 2 + bc]); \
                    const double* zq = zv + MF_TS * ((kb) - W1_C + c) + lg;                                          \
                    part[bi] += (cy[c][bi][0] * zq[0] + cy[c][bi][1] * zq[4]) + (cy[c][bi][2] * zq[8] + cy[c][bi][3] * zq[12]); \
                }                                                                                                    \
            }                                                                                                        \
        } else {     /* (zeroed HERE, not up front: ten live zero tiles beside the sixteen carry tiles do not fit) */       \
            _Pragma("unroll") for (int bc = 0; bc <= (bi); ++bc) tacc[(bi) * ((bi) + 1) / 2 + bc] = d4{0.0, 0.0, 0.0, 0.0}; \
        }                                                                                                            \
    } while (0)
#define W1_BLOCK_INIT(kb, nb, WITH_CY)                                                                               \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < W1_C; ++i) part[i] = 0.0;                                              \
        /* (scheduling barriers: left alone, hipcc evaluates all ten Gram tiles first -- 80 registers beside the sixteen carry   \
           tiles -- and spills the accumulators) */                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        W1_BLOCK_ROW(3, kb, nb, WITH_CY);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        W1_BLOCK_ROW(2, kb, nb, WITH_CY);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        W1_BLOCK_ROW(1, kb, nb, WITH_CY);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        W1_BLOCK_ROW(0, kb, nb, WITH_CY);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#if W1_CARRY
        {
            d4 cy[W1_C][4];       // (never read: step 0 has no columns before it)
            const int nb0 = min(W1_C, nt);
            W1_BLOCK_INIT(0, nb0, false);
        }
#endif
        const bool lastres = W1_LASTRES && (nt & 3) == 0 && !g.export_factor;
        W1_STAMP(1);
        for (int k = 0;; k += W1_C) {                                      // (left by `break` at the last step)
            const int nc = min(W1_C, nt - k);                              // tile columns of this step
            const int jend = (W1_CARRY && k > 0) ? k - W1_C : k;           // the sweep reads tile columns j < jend from the workspace
            const int kl = jend - 1;
            const bool last_step = k + W1_C >= nt;
            W1_FRESH_LANE();
            // ---- the diagonal block: T_(k+i)(k+c) = A - sum_{j<k} L_(k+i)j L_(k+c)j^T, tile d = i (i + 1) / 2 + c, one sweep over j;
            //      the forward-solve sums  part_c = sum_{j<k} L_(k+c)j z_j  from the same operands ----
            const double* rrow[W1_C];
#pragma unroll
            for (int i = 0; i < W1_C; ++i) rrow[i] = Lt + W1_TILE(k + min(i, nc - 1), 0);
#if !W1_CARRY
            d4 tacc[W1_NDT];
            double part[W1_C];
#endif
            d4 op[2][W1_C];
#pragma unroll
            for (int i = 0; i < W1_C; ++i) op[0][i] = op[1][i] = d4{0.0, 0.0, 0.0, 0.0};
            if (jend > 0) {
#pragma unroll
                for (int i = 0; i < W1_C; ++i) op[0][i] = mf_img_load(rrow[i], lane);
            }
#if !W1_CARRY
            {
                d4 cy[W1_C][4];   // (never read)
                W1_BLOCK_INIT(k, nc, false);
            }
#endif
            for (int j = 0; j < jend; j += 2) {
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
            // (W1_LASTRES: the last step's block never reaches the workspace -- its tiles go into the backward solve as registers, its
            // L_cc^-T images are read back transposed from the L_cc^-1 images in LDS)
            const bool keep = lastres && last_step;
            // (W1_LASTRES_NOSTORE: the block's tiles and L_cc^-T are not even written -- 10 image writes less, but the conditional stores
            // cost the chain 100 spilled VGPRs; by default only the READS of the backward solve go)
#ifndef W1_LASTRES_NOSTORE
#define W1_LASTRES_NOSTORE 0
#endif
            const bool skip_store = W1_LASTRES_NOSTORE && keep;
#if W1_CMASK
#define W1_DIAG(Wt, c_) mf_diag_factor_c(Wt, rsbuf, LinvC + (c_) * MF_IMG, skip_store ? nullptr : LinvTg + (size_t)(k + (c_)) * MF_IMG, ptol, lane)
#else
#define W1_DIAG(Wt, c_) mf_diag_factor<true>(Wt, rsbuf, LinvC + (c_) * MF_IMG, LinvTg + (size_t)(k + (c_)) * MF_IMG, ptol, lane)
#endif
            bool ok = W1_DIAG(tacc[0], 0);
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
                        if (!skip_store) W1_STORE_TILE(Lt + W1_TILE(k + i, k + c), lane, L);
                    }
                    d4 Dii = tacc[i * (i + 1) / 2 + i];
#pragma unroll
                    for (int c = 0; c < i; ++c) Dii = w1_mfma4_neg(Lb[i * (i - 1) / 2 + c], Lb[i * (i - 1) / 2 + c], Dii);
                    ok = W1_DIAG(Dii, i);
                    W1_LDS_SYNC();
                    if (g.export_factor) mf_img_store(Lt + W1_TILE(k + i, k + i), lane, mf_img_load(LinvC + i * MF_IMG, mf_opaque(lane)));
                }
            }
#undef W1_DIAG
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
            const int rows_tot = nt - (k + nc);
            if (rows_tot == 0) {
                // the last step: no rows below its block.  (The loop's only regular exit: values kept for the backward solve are live on
                // this edge alone, not across the row passes of the earlier steps.)
                if (keep) {
#if !defined(W1_EXP_NOBACK)
                    // W1_LASTRES: the last four columns of the backward solve right here, from the block's tiles in registers and the
                    // L_cc^-1 images in LDS read transposed (tile (k + 1 + t, k) of column kk from the end is block tile
                    // (4 - kk + t, 3 - kk)) -- nothing of the last block is written to or read from the workspace, and nothing of it
                    // is live beyond this branch
                    W1_FRESH_LANE();
                    d4 pa0 = d4{0.0, 0.0, 0.0, 0.0};
                    w1_static_for<0, NLAST>([&](auto P) __attribute__((always_inline)) {
                        constexpr int q = decltype(P)::value;
                        constexpr int kk = w1_stream_col(q), t = q - kk * (kk + 1) / 2;
                        d4 img;
                        if constexpr (t < kk) {
                            constexpr int bi = W1_C - kk + t, bc = W1_C - 1 - kk;
                            img = Lb[bi * (bi - 1) / 2 + bc];
                        } else {
                            const double* Mi = LinvC + (W1_C - 1 - kk) * MF_IMG;
#pragma unroll
                            for (int s_ = 0; s_ < 4; ++s_) img[s_] = Mi[mf_img_rc(lg + 4 * s_, lr)];
                        }
                        bw_step(P, img, pa0);
                    });
#endif
                } else {
                    __syncthreads();   // the block's tiles are in the workspace before the backward solve reads them back
                }
                break;
            }
            if (k > 0) __syncthreads();     // the chain's tiles are in the workspace before the row passes read them back
            // ---- rows k + 4 .. nt - 1 (rows below a block exist only when the block has all four columns), FOUR per pass: update the
            //      sixteen accumulators over j < k, then column by column  T_r(k+c) -= sum_{c2<c} L_r(k+c2) L_(k+c)(k+c2)^T,
            //      L_r(k+c) = T L_cc^-T.  The pass of the NEXT block's rows (first_row 0) runs LAST and its result tiles stay in cy for
            //      the next step's block (W1_CARRY). ----
            const int npass = (rows_tot + 3) / 4;      // (a descending loop makes hipcc spill 90 more registers: the passes run 4, 8, .., then 0)
            const int klp = k - 1;
#if W1_CARRY
            d4 cy[W1_C][4];                            // the accumulators of a pass; after the loops: the tiles of the pass that ran last
#endif
            if (k == 0) {
                // Step 0 has no update loop, so its registers are free: the block's lower tiles stay resident (no reload from the
                // workspace -- at eight patches per CU those reloads miss L2) and four independent TRSM chains interleave.
                for (int pi_ = 0; pi_ < npass; ++pi_) {
                    const int first_row0 = (pi_ + 1 < npass) ? 4 * (pi_ + 1) : 0;
                    const int np4 = min(4, rows_tot - first_row0);
                    W1_FRESH_LANE();
#if !W1_CARRY
                    d4 cy[W1_C][4];
#endif
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r_ = W1_C + first_row0 + min(t, np4 - 1);       // (t >= np4: a copy of the last row, never stored)
                        d4 gv[W1_C];
                        W1_GRAM_ROW(gv, 2, false, r_, 0);
                        cy[0][t] = gv[0]; cy[1][t] = gv[1];
                        __builtin_amdgcn_sched_barrier(0);
                        W1_GRAM_ROW(gv, 2, false, r_, 2);
                        cy[2][t] = gv[0]; cy[3][t] = gv[1];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    W1_STAMP(5);
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) {
                        const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
#pragma unroll
                            for (int c2 = 0; c2 < c; ++c2) cy[c][t] = w1_mfma4_neg(Lb[c * (c - 1) / 2 + c2], cy[c2][t], cy[c][t]);
                            cy[c][t] = w1_trsm(lv, cy[c][t]);
                            if (t < np4) W1_STORE_TILE(Lt + W1_TILE(W1_C + first_row0 + t, c), lane, cy[c][t]);
                        }
                    }
                    W1_STAMP(7);
                }
            }
            if (k > 0) {    // (a second `if`, not an `else`: hipcc lays an else-branch out FIRST and keeps the block tiles Lb, which only step 0's
                            // passes read, live across it -- five tiles spilled and reloaded around the passes of every later step)
                // k > 0: 16 accumulators; the four column operands L_(k+c)j double-buffered, the four row operands L_rj single-buffered and
                // re-requested for j + 1 as soon as their four products of j are issued (twelve products of lead).  Eight images per 16
                // products: the two-row passes of the first version fetched six per eight, and at eight patches per CU every one of them
                // is an L2 miss (the kernel runs at the HBM bandwidth the part delivers).
                for (int pi_ = 0; pi_ < npass; ++pi_) {
                    const int first_row = (pi_ + 1 < npass) ? 4 * (pi_ + 1) : 0;
                    const int np4 = min(4, rows_tot - first_row);
                    W1_FRESH_LANE();
#if !W1_CARRY
                    d4 cy[W1_C][4];
#endif
                    int rr[4];
                    const double* rw_[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        rr[t] = k + W1_C + first_row + min(t, np4 - 1);        // (t >= np4: a copy of the last row, never stored)
                        rw_[t] = Lt + W1_TILE(rr[t], 0);
                    }
                    d4 A2[2][W1_C], B[4];
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) A2[0][c] = mf_img_load(rrow[c], lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t) B[t] = W1_LOAD_ROWOP(rw_[t], lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        d4 gv[W1_C];
                        W1_GRAM_ROW(gv, 2, false, rr[t], k);
                        cy[0][t] = gv[0]; cy[1][t] = gv[1];
                        __builtin_amdgcn_sched_barrier(0);
                        W1_GRAM_ROW(gv, 2, false, rr[t], k + 2);
                        cy[2][t] = gv[0]; cy[3][t] = gv[1];
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
            _Pragma("unroll") for (int c = 0; c < W1_C; ++c) cy[c][t] = w1_mfma4_neg(A2[st][c], B[t], cy[c][t]);      \
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
                    W1_PASS_J(0, klp, true);                // (k is a multiple of four: the last pair of j; its second half requests nothing)
                    W1_PASS_J(1, klp, false);
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
                            for (int c2 = 0; c2 < c; ++c2) cy[c][t] = w1_mfma4_neg(Lq[c * (c - 1) / 2 + c2], cy[c2][t], cy[c][t]);
                            cy[c][t] = w1_trsm(lv, cy[c][t]);
#endif
                            if (t < np4) W1_STORE_TILE(Lt + W1_TILE(rr[t], k + c), lane, cy[c][t]);
                        }
                    }
                    W1_STAMP(7);
                }
            }
#if W1_CARRY
            // ---- the next step's block, while the tiles of its rows over this step's columns are in registers ----
            W1_FRESH_LANE();
            {
                const int kn = k + W1_C, nbn = min(W1_C, nt - kn);
                W1_BLOCK_INIT(kn, nbn, true);
            }
#endif
            __syncthreads();   // the column block is in the workspace (this wave's own stores, read back by its next sweep and passes)
            W1_STAMP(8);
        }
#undef W1_BLOCK_INIT
#undef W1_BLOCK_ROW

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
            // W1_LASTRES: the first NLAST positions -- the last block -- were consumed where the factorization ended; the stream starts
            // behind them
            const int bw_start = lastres ? NLAST : 0;
            w1_static_for<0, NLAST + W1_BW>([&](auto P) __attribute__((always_inline)) {
                constexpr int q = decltype(P)::value;
                if constexpr (q < SLEN) {
                    if (q >= bw_start && q < bw_start + W1_BW) win[q % W1_BW] = W1_LOAD_BACK(stream_addr(P), lane);
                }
            });
            d4 pa = d4{0.0, 0.0, 0.0, 0.0};
            w1_static_for<0, SLEN>([&](auto P) __attribute__((always_inline)) {
                constexpr int q = decltype(P)::value;
                if (q >= NLAST || q >= bw_start) {
                    bw_step(P, win[q % W1_BW], pa);
                    if constexpr (q + W1_BW < SLEN) {
                        win[q % W1_BW] = W1_LOAD_BACK(stream_addr(std::integral_constant<int, q + W1_BW>{}), lane);
                    }
                }
            });
        }
#endif
        W1_STAMP(9);
        double* av = zv;
        if (A.alpha_out)
            for (int i = lane; i < n; i += 64) A.alpha_out[o + i] = av[i] * g.a_out;      // (unit scale: the weights of K are a' / sigma_f^2)

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
            double gqx[2], gqy[2];        // grid coordinates of this lane's rows / columns in the coordinates the points are held in
            const double cs = scaled ? g.cs : 1.0, co0 = scaled ? A.x0[o] : 0.0, co1 = scaled ? A.x1[o] : 0.0;
            const double argc = scaled ? -1.0 : cexp, sfp = g.sfp;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double gq = res * (((double)(16 * h + lr) + 0.5) / (double)sz - 0.5);
                gqx[h] = cs * (gq - co0);
                gqy[h] = cs * (gq - co1);
            }
            // No masks in the loop: a point beyond n contributes nothing because its weight is SELECTED to zero (alpha is zero from n
            // to 16 nt -- identity padding solves to 0 -- but never written beyond: select, do not multiply), and the grid rows / columns
            // beyond sz are finite numbers that are never stored.  One basic block per 32 points: 32 independent exponentials for the
            // scheduler (behind per-element masks and mode branches this phase took a sixth of the kernel).
            auto predict_loop = [&](auto mode_tag) __attribute__((always_inline)) {
                constexpr int MP = decltype(mode_tag)::value;
                for (int ibase = 0; ibase < n; ibase += 32) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ibase + 4 * s + lg;
                        const int ic = min(i, W1_NPAD - 1);
                        const double al = (i < n) ? sfp * av[ic] : 0.0;
                        const double xi0 = px0[ic], xi1 = px1[ic];
                        double ea[2], eb[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double dy = gqy[h] - xi1, dx = gqx[h] - xi0;
                            if constexpr (MP == 0) {
                                ea[h] = gpc_exp_neg(argc * (dy * dy), T);                 // Ey[py][i]
                                eb[h] = gpc_exp_neg(argc * (dx * dx), T) * al;            // Ex[px][i] * weight_i
                            } else {
                                ea[h] = gpc_expm_poly<MP == 1 ? 7 : 5>(dy * dy);
                                eb[h] = gpc_expm_poly<MP == 1 ? 7 : 5>(dx * dx) * al;
                            }
                        }
#pragma unroll
                        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                P[mt][nl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[mt], eb[nl], P[mt][nl], 0, 0, 0);
                    }
                }
            };
            if (mode_p == 2) predict_loop(std::integral_constant<int, 2>{});
            else if (mode_p == 1) predict_loop(std::integral_constant<int, 1>{});
            else predict_loop(std::integral_constant<int, 0>{});
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
            const double cs = scaled ? g.cs : 1.0, co0 = scaled ? A.x0[o] : 0.0, co1 = scaled ? A.x1[o] : 0.0;
            const double argc = scaled ? -1.0 : cexp, sfp = g.sfp;
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
                q0 = cs * (q0 - co0);
                q1 = cs * (q1 - co1);
                double s_ = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double d0 = px0[i] - q0, d1 = px1[i] - q1;
                    s_ += gpc_exp_neg(argc * __builtin_fma(d0, d0, d1 * d1), T) * av[i];
                }
                fs[p] = sfp * s_;
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

static int w1_npad(const DenseArgs& a) { return a.n_max <= 256 ? 256 : W1_NPAD_MAX; }

bool dense_w1_supported(const DenseArgs& a)
{
    // (with the variance: point-wise X* only -- the variance entry has no grid form -- and n <= 256: the 512-point instance's slots
    // are not the layout dense_variance_big_kernel reads)
    if (a.v_star) return a.n_max <= 256 && a.ny == 1 && !a.sel && a.xs0 != nullptr;
    return a.n_max <= W1_NPAD_MAX && a.ny == 1 && !a.sel;
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
    const int npad = w1_npad(a);
    return sizeof(double) * ((size_t)(W1_TRI_OF(npad) + W1_NT_OF(npad)) * MF_IMG * (size_t)grid + (a.v_star ? (size_t)a.n_total : 0));
}

int dense_w1_launch(gpc_ctx* ctx, const DenseArgs& a_in, int grid)
{
    DenseArgs a = a_in;
    double* v_star = a.v_star;
    a.v_star = nullptr;
    W1Params g;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.ws = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + ctx->ws_off);
    const int npad = w1_npad(a);
    g.linvt = g.ws + (size_t)W1_TRI_OF(npad) * MF_IMG * (size_t)grid;
    g.export_factor = v_star ? 1 : 0;
    const double sf = a.prm.sigmaf_sq;
    const bool unit = !v_star;                        // unit scale (see the kernel) unless the factor is exported
    g.cs = sqrt(-g.c_exp);
    g.noise_u = unit ? a.prm.noise / sf : a.prm.noise;
    g.pivot_tol = GPC_PIVOT_RTOL * (unit ? 1.0 + a.prm.noise / sf : sf + a.prm.noise);
    g.sfp = unit ? 1.0 : sf;
    g.a_out = unit ? 1.0 / sf : 1.0;
    g.stamps = nullptr;
    if (v_star) {
        // Predictive variance (gaussian_process::predict_measurements, /root/reference/src/gaussian_process.cpp:35-43): the slots are the
        // factor export dense_variance_kernel<16> reads, the fit predicts nothing (the solve forms the mean from the same K* tiles)
        a.m = 0;
        if (!a.alpha_out) a.alpha_out = g.linvt + (size_t)W1_NT_OF(npad) * MF_IMG * (size_t)grid;
    }
    ctx->last_dense_kernel = v_star ? "dense_mfma_w1 + dense_variance" : npad == 256 ? "dense_mfma_w1" : "dense_mfma_w1_512";
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
        if (npad == 256) hipLaunchKernelGGL((dense_w1_kernel<256>), dim3(cnt), dim3(64), pad, ctx->stream, g);
        else hipLaunchKernelGGL((dense_w1_kernel<W1_NPAD_MAX>), dim3(cnt), dim3(64), pad, ctx->stream, g);
        GPC_HIP(ctx, hipGetLastError());
        if (v_star) {
            DenseArgs av = g.a;
            av.m = a_in.m;
            const int rc = dense_variance_launch(ctx, av, W1_NT_OF(256), g.ws, a.alpha_out, v_star + (size_t)base * a_in.m);
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
