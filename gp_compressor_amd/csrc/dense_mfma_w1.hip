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

#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"

#define W1_NPAD 256
#define W1_C 4          // tile columns per step
#define W1_NDT 10       // tiles of a step's diagonal block

struct W1Params {
    DenseArgs a;
    double c_exp;
    double pivot_tol;
    double* ws;
    size_t slot;        // doubles per slot (big_slot_doubles: same layout as the tiled kernel's)
    int ntw;            // tile columns of a slot
    int export_factor;  // slot = patch (the factor of every patch stays, with the L_kk^-1 images): predictive variance
};

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

// Gram tile (tile row r, tile column c) in the transposed C/D layout: register q of lane l = K[16 r + (l & 15)][16 c + (l >> 4) + 4 q]
// (+ the noise diagonal, identity padding beyond n).  r, c, the mode and the two special cases are wave-uniform: interior
// off-diagonal tiles -- nine in ten -- take the bare path (distance, exponential, nothing else).
template <bool SMALL>
__device__ static __forceinline__ d4 w1_gram_tile(const double* px0, const double* px1, const double* T, double sf, double cexp, double noise,
                                                  bool dbl, int n, int r, int c, int lr, int lg)
{
    const int pi = MF_TS * r + lr;
    const double xi0 = px0[pi], xi1 = px1[pi];
    d4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pj = MF_TS * c + lg + 4 * q;
        v[q] = SMALL ? gpc_rbf_small(sf, cexp, xi0, xi1, px0[pj], px1[pj]) : gpc_rbf_neg(sf, cexp, xi0, xi1, px0[pj], px1[pj], T);
    }
    if (r == c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pj = MF_TS * c + lg + 4 * q;
            if (pi == pj) {
                v[q] += noise;               // covariance_matrix(..., training)   gaussian_process.cpp:59-61
                if (dbl) v[q] += noise;      // C.diagonal() += sigman_sq          :21
            }
        }
    }
    if (MF_TS * (r + 1) > n || MF_TS * (c + 1) > n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pj = MF_TS * c + lg + 4 * q;
            if (pi >= n || pj >= n) v[q] = (pi == pj) ? 1.0 : 0.0;      // identity padding
        }
    }
    return v;
}

__global__ __launch_bounds__(64, 2) void dense_w1_kernel(W1Params g)
{
    __shared__ __attribute__((aligned(16))) double T[GPC_EXP_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) double px0[W1_NPAD], px1[W1_NPAD], zv[W1_NPAD];     // zv: z, then alpha in place
    __shared__ __attribute__((aligned(16))) double rsbuf[32], wsc[32];
    __shared__ __attribute__((aligned(16))) double LinvC[W1_C * MF_IMG];

    const DenseArgs& A = g.a;
    const int lane = threadIdx.x;
    const int lr = lane & 15, lg = lane >> 4;
    const int m = A.m, ntw = g.ntw;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    const bool dbl = A.prm.ref_double_noise != 0;
    double* Lt = g.ws + (size_t)blockIdx.x * g.slot;              // tiles (i, j): Lt + (i * ntw + j) * 256
    double* LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;       // L_kk^-T images
    double* LinvG = LinvTg + (size_t)ntw * MF_IMG;                // L_kk^-1 images (export only)

    gpc_exp_table_init(T);

    const int n_patches = A.sel ? __builtin_amdgcn_readfirstlane(A.sel_count[0]) : A.P;
    for (int pk = blockIdx.x; pk < n_patches; pk += gridDim.x) {
        const int patch = A.sel ? __builtin_amdgcn_readfirstlane(A.sel[pk]) : pk;
        if (g.export_factor) {
            Lt = g.ws + (size_t)patch * g.slot;
            LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;
            LinvG = LinvTg + (size_t)ntw * MF_IMG;
        }
        const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
        const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
        double* fs = A.f_star + (size_t)patch * m;
        __syncthreads();   // (one wave: a fence -- the previous patch is done with LDS)
        if (n <= 0 || n > MF_TS * ntw || n > W1_NPAD) {
            for (int p = lane; p < m; p += 64) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (lane == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            continue;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);

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
            zv[i] = 0.0;
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
        __syncthreads();
        auto gram = [&](int r, int c) __attribute__((always_inline)) {
            return small_gram ? w1_gram_tile<true>(px0, px1, T, sf, cexp, noise, dbl, n, r, c, lr, lg)
                              : w1_gram_tile<false>(px0, px1, T, sf, cexp, noise, dbl, n, r, c, lr, lg);
        };

        bool bad = false;
        // ---- tiled left-looking Cholesky, four tile columns (k .. k+3) per step ----
        for (int k = 0; k < nt; k += W1_C) {
            const int nc = min(W1_C, nt - k);                              // tile columns of this step
            const int kl = k - 1;
            // ---- the diagonal block: T_(k+i)(k+c) = A - sum_{j<k} L_(k+i)j L_(k+c)j^T, tile d = i (i + 1) / 2 + c, one sweep over j;
            //      the forward-solve sums  part_c = sum_{j<k} L_(k+c)j z_j  from the same operands ----
            const double* rrow[W1_C];
#pragma unroll
            for (int i = 0; i < W1_C; ++i) rrow[i] = Lt + ((size_t)(k + min(i, nc - 1)) * ntw) * MF_IMG;
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
            for (int d = 0; d < W1_NDT; ++d) {
                const int bi = d >= 6 ? 3 : d >= 3 ? 2 : d >= 1 ? 1 : 0, bc = d - bi * (bi + 1) / 2;
                tacc[d] = d4{0.0, 0.0, 0.0, 0.0};
                if (bi < nc) tacc[d] = gram(k + bi, k + bc);
            }
            for (int j = 0; j < k; j += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int jn = min(j + h + 1, kl);
#pragma unroll
                    for (int i = 0; i < W1_C; ++i) op[h ^ 1][i] = mf_img_load(rrow[i] + (size_t)jn * MF_IMG, lane);
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
            // ---- the chain, in registers: factor (k,k); then row by row  L_ic = (T_ic - sum_{c2<c} L_ic2 L_cc2^T) L_cc^-T,
            //      T_ii -= sum_c L_ic L_ic^T, factor.  Lb[i (i-1)/2 + c] = operand image of L_(k+i)(k+c), c < i. ----
            d4 Lb[W1_C * (W1_C - 1) / 2];
#pragma unroll
            for (int q = 0; q < W1_C * (W1_C - 1) / 2; ++q) Lb[q] = d4{0.0, 0.0, 0.0, 0.0};
            bool ok = mf_diag_factor<true>(tacc[0], rsbuf, LinvC, LinvTg + (size_t)k * MF_IMG, g.pivot_tol);
            __syncthreads();
            if (g.export_factor) mf_img_store(LinvG + (size_t)k * MF_IMG, lane, mf_img_load(LinvC, lane));
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
                        mf_img_store(Lt + ((size_t)(k + i) * ntw + k + c) * MF_IMG, lane, L);
                    }
                    d4 Dii = tacc[i * (i + 1) / 2 + i];
#pragma unroll
                    for (int c = 0; c < i; ++c) Dii = w1_mfma4_neg(Lb[i * (i - 1) / 2 + c], Lb[i * (i - 1) / 2 + c], Dii);
                    ok = mf_diag_factor<true>(Dii, rsbuf, LinvC + i * MF_IMG, LinvTg + (size_t)(k + i) * MF_IMG, g.pivot_tol);
                    __syncthreads();
                    if (g.export_factor) mf_img_store(LinvG + (size_t)(k + i) * MF_IMG, lane, mf_img_load(LinvC + i * MF_IMG, mf_opaque(lane)));
                }
            }
            if (!ok) { bad = true; break; }
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
                    const double yv = (pj < n) ? A.y[o + pj] : 0.0;
                    if (lg == 0) zv[pj] = yv - pp;            // t_c, where z_c goes next
                    __syncthreads();
                    const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
                    const double* tq = zv + MF_TS * (k + c) + lg;
                    double zz = (lv[0] * tq[0] + lv[1] * tq[4]) + (lv[2] * tq[8] + lv[3] * tq[12]);
                    zz += __shfl_xor(zz, 16, 64);
                    zz += __shfl_xor(zz, 32, 64);
                    __syncthreads();
                    if (lg == 0) zv[pj] = zz;
                    __syncthreads();
                }
            }
            // ---- rows k + nc .. nt - 1, two per pass: update the four accumulators over j < k, then column by column
            //      T_r(k+c) -= sum_{c2<c} L_r(k+c2) L_(k+c)(k+c2)^T,  L_r(k+c) = T L_cc^-T ----
            const int rows_tot = nt - (k + nc);
            for (int first_row = 0; first_row < rows_tot; first_row += 2) {
                const int np = min(2, rows_tot - first_row);
                d4 acc[W1_C][2];
                int rr[2];
                const double* rw_[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    rr[t] = k + nc + first_row + min(t, np - 1);           // (t >= np: a copy of the last row, never stored)
                    rw_[t] = Lt + ((size_t)rr[t] * ntw) * MF_IMG;
                }
                d4 sa[2][W1_C], sb[2][2];
                // two operand stages; the loads are UNCONDITIONAL (indices clamped: redundant re-reads) so that hipcc can count the
                // outstanding ones; the first stage is requested before the Gram tiles are evaluated
#define W1_LOAD_STAGE(st, jj, NPC)                                                                                   \
    do {                                                                                                             \
        _Pragma("unroll") for (int c = 0; c < W1_C; ++c) sa[st][c] = mf_img_load(rrow[c] + (size_t)(jj) * MF_IMG, lane); \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t) sb[st][t] = mf_img_load(rw_[t] + (size_t)(jj) * MF_IMG, lane);  \
    } while (0)
#define W1_USE_STAGE(st, NPC)                                                                                        \
    do {                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                              \
            _Pragma("unroll") for (int c = 0; c < W1_C; ++c) acc[c][t] = w1_mfma4_neg(sa[st][c], sb[st][t], acc[c][t]); \
    } while (0)
#define W1_UPDATE_LOOP(NPC)                                                                                          \
    do {                                                                                                             \
        if (k > 0) W1_LOAD_STAGE(0, 0, NPC);                                                                         \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                              \
            _Pragma("unroll") for (int c = 0; c < W1_C; ++c)                                                         \
                if (c < nc) acc[c][t] = gram(rr[t], k + c);                                                          \
        for (int j = 0; j < k; j += 2) {                                                                             \
            W1_LOAD_STAGE(1, j + 1, NPC);                                                                            \
            W1_USE_STAGE(0, NPC);                                                                                    \
            W1_LOAD_STAGE(0, min(j + 2, kl), NPC);                                                                   \
            W1_USE_STAGE(1, NPC);                                                                                    \
        }                                                                                                            \
    } while (0)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) acc[c][t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int st = 0; st < 2; ++st) {
#pragma unroll
                    for (int c = 0; c < W1_C; ++c) sa[st][c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int t = 0; t < 2; ++t) sb[st][t] = d4{0.0, 0.0, 0.0, 0.0};
                }
                if (np == 2) W1_UPDATE_LOOP(2);
                else W1_UPDATE_LOOP(1);
#pragma unroll
                for (int c = 0; c < W1_C; ++c) {
                    if (c < nc) {
                        const d4 lv = mf_img_load(LinvC + c * MF_IMG, mf_opaque(lane));
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if (t < np) {
#pragma unroll
                                for (int c2 = 0; c2 < c; ++c2) acc[c][t] = w1_mfma4_neg(Lb[c * (c - 1) / 2 + c2], acc[c2][t], acc[c][t]);
                                acc[c][t] = w1_trsm(lv, acc[c][t]);
                                mf_img_store(Lt + ((size_t)rr[t] * ntw + k + c) * MF_IMG, lane, acc[c][t]);
                            }
                        }
                    }
                }
            }
            __syncthreads();   // the column block is in the workspace (this wave's own stores, read back by its next sweep)
        }

        if (bad) {
            for (int p = lane; p < m; p += 64) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = lane; i < n; i += 64) A.alpha_out[o + i] = __builtin_nan("");
            if (lane == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            continue;
        }

        // ---- backward solve L^T alpha = z, tile columns from the last to the first; alpha replaces z in place ----
        // Column k: w_k = sum_{i>k} L_ik^T alpha_i on the VALU (the products contract over the ROW index, which the image layout
        // cannot feed to an MFMA) with the transposing DPP row reduction, then alpha_k = L_kk^-T (z_k - w_k): four MFMAs.
        // The tiles do not depend on alpha: rows k+1 .. k+8 of column k are requested one column ahead, the rest at its start.
        {
            d4 cur[8], nxt[8], lt_cur, lt_nxt;
#pragma unroll
            for (int t = 0; t < 8; ++t) cur[t] = nxt[t] = d4{0.0, 0.0, 0.0, 0.0};
            lt_cur = mf_img_load(LinvTg + (size_t)(nt - 1) * MF_IMG, lane);
            lt_nxt = lt_cur;
            for (int k = nt - 1; k >= 0; --k) {
                d4 far[7];
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    const int i = k + 9 + t;
                    far[t] = d4{0.0, 0.0, 0.0, 0.0};
                    if (i < nt) far[t] = mf_img_load(Lt + ((size_t)i * ntw + k) * MF_IMG, lane);
                }
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int i = k + t;                                  // rows of column k-1: i >= k
                        if (i < nt) nxt[t] = mf_img_load(Lt + ((size_t)i * ntw + (k - 1)) * MF_IMG, lane);
                    }
                    lt_nxt = mf_img_load(LinvTg + (size_t)(k - 1) * MF_IMG, lane);
                }
                d4 pa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = k + 1 + t;
                    if (i < nt) pa += cur[t] * zv[MF_TS * i + lr];          // cur[t] = L_ik[l & 15][(l >> 4) + 4 s]
                }
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    const int i = k + 9 + t;
                    if (i < nt) pa += far[t] * zv[MF_TS * i + lr];
                }
                d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                if (k + 1 < nt) {
                    const double tot = mf_row_reduce4(pa, lr);               // lanes lr = 0, 4, 8, 12 hold components 0 .. 3
                    if ((lr & 3) == 0) wsc[lg + 4 * (lr >> 2)] = tot;
                    __syncthreads();
                    if (lr == 0) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) ub[q4] = zv[MF_TS * k + lg + 4 * q4] - wsc[lg + 4 * q4];
                    }
                } else if (lr == 0) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) ub[q4] = zv[MF_TS * k + lg + 4 * q4];
                }
                const d4 al = w1_trsm(lt_cur, ub);                           // lanes lr = 0: alpha[16 k + (l >> 4) + 4 r]
                __syncthreads();
                if (lr == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) zv[MF_TS * k + lg + 4 * r] = al[r];
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
                lt_cur = lt_nxt;
            }
        }
        double* av = zv;
        if (A.alpha_out)
            for (int i = lane; i < n; i += 64) A.alpha_out[o + i] = av[i];

        // ---- predictive mean ----
        if (A.xs0 == nullptr && A.grid_sz <= 32) {
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
            for (int ibase = 0; ibase < n; ibase += 32) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i = ibase + 4 * s + lg;
                    // alpha is zero from n to 16 nt (identity padding solves to 0) but never written beyond: select, do not multiply
                    const int ic = min(i, W1_NPAD - 1);
                    const double al = (i < n) ? sf * av[ic] : 0.0;
                    const double xi0 = px0[ic], xi1 = px1[ic];
                    double ea[2], eb[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bool on = (16 * h + lr < sz) && (i < n);
                        const double dy = gq[h] - xi1, dx = gq[h] - xi0;
                        if (small_grid) {
                            ea[h] = on ? gpc_exp_small(cexp * (dy * dy)) : 0.0;         // Ey[py][i]
                            eb[h] = on ? gpc_exp_small(cexp * (dx * dx)) * al : 0.0;    // Ex[px][i] * sf alpha_i
                        } else {
                            ea[h] = on ? gpc_exp_neg(cexp * (dy * dy), T) : 0.0;
                            eb[h] = on ? gpc_exp_neg(cexp * (dx * dx), T) * al : 0.0;
                        }
                    }
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            P[mt][nl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[mt], eb[nl], P[mt][nl], 0, 0, 0);
                }
            }
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
        if (lane == 0 && A.status) A.status[patch] = GPC_STATUS_OK;
    }
}

bool dense_w1_supported(const DenseArgs& a)
{
    return a.n_max <= W1_NPAD && a.ny == 1 && !a.v_star;
}

static int w1_per_cu() { const char* e = getenv("GPC_W1_PER_CU"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 8 ? v : 8; }

size_t dense_w1_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out)
{
    const int ntw = (a.n_max + MF_TS - 1) / MF_TS;
    const int cap = ctx->num_cus * w1_per_cu();
    const int grid = a.P < cap ? a.P : cap;
    if (grid_out) *grid_out = grid;
    return sizeof(double) * big_slot_doubles(ntw) * (size_t)grid;
}

int dense_w1_launch(gpc_ctx* ctx, const DenseArgs& a, int grid)
{
    W1Params g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.ws = static_cast<double*>(ctx->ws);
    g.ntw = (a.n_max + MF_TS - 1) / MF_TS;
    g.slot = big_slot_doubles(g.ntw);
    g.export_factor = 0;
    ctx->last_dense_kernel = "dense_mfma_w1";
    hipLaunchKernelGGL(dense_w1_kernel, dim3(grid), dim3(64), 0, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}
