// dense_mfma_big.hip -- batched dense GP for 256 < n <= 1024 (BASELINE config 3: n = 512, config 5: n = 1024) on gfx950.
//
// Same computation as dense_mfma.hip / dense_generic.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45).  At these sizes the lower triangle of K (1 MB at n = 512, 4 MB at
// n = 1024) exceeds the register file of a CU, so the factor lives in a per-workgroup slot of a global workspace as
// 16 x 16 tiles in MFMA *operand image* layout (mfma_tile.h; 2 KB each, read back with two coalesced 16-byte loads per
// lane) and the factorisation is a tiled LEFT-looking Cholesky:
//
//   column k:  T_rk = A_rk - sum_{j<k} L_rj L_kj^T   (4 v_mfma_f64_16x16x4_f64 per (r, j)),   r = k .. nt
//              L_kk^-1 by the in-place Gauss-Jordan on the MFMA pipe (mf_diag_factor),   L_rk = T_rk L_kk^-T (4 MFMAs)
//
// * K itself is never stored: A_rk is evaluated (RBF + noise diagonal) straight into the accumulator registers when
//   column k starts.  HBM/L2 see the factor only: one write per tile, nt/3 reads on average.
// * Tile rows of a column are dealt round-robin to the 8 waves; a wave keeps up to BG_RMAX row accumulators and loads
//   the shared operand L_kj once per j for all of them (register blocking), with the next j prefetched.
// * The right-hand sides ride along as one more tile row (row nt holds y^T padded to 16 channels): its TRSM result is
//   z^T, so the forward solve needs no code of its own.
// * Wave 0 takes the diagonal tile first and factors it while the other waves still stream their updates; they poll an
//   LDS word for L_kk^-1 (bounded asm poll, mfma_tile.h).  One workgroup barrier per column.
// * Backward solve: alpha_k = L_kk^-T (z_k - sum_{i>k} L_ik^T alpha_i); the tile products contract over the ROW index,
//   which the image layout cannot feed to an MFMA, so they run on the VALU with the DPP row reduction of mfma_tile.h
//   (O(n^2) work against the O(n^3) of the factorisation).  Predictive mean: the separable-grid MFMA form of
//   dense_mfma.hip, looped over 32-point chunks.
#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"

#define BG_THREADS 512
#define BG_WAVES 8
#define BG_NPAD 1024
#define BG_RMAX 5   // tile rows a wave updates per pass (5 x 8 accumulator + 2 x 5 x 8 operand VGPRs)

struct BigParams {
    DenseArgs a;
    double c_exp;
    double pivot_tol;
    double* ws;
    size_t slot;   // doubles per workgroup slot
    int ntw;       // tile columns of a slot = ceil(n_max / 16)
};

// LDS carve (doubles)
#define B_EXP 0
#define B_PX0 64
#define B_PX1 (B_PX0 + BG_NPAD)
#define B_ZV (B_PX1 + BG_NPAD)
#define B_WV (B_ZV + 3 * BG_NPAD)
#define B_AV (B_WV + 3 * BG_NPAD)
#define B_RS (B_AV + 3 * BG_NPAD)      // 32: rsqrt row of mf_diag_factor
#define B_FLAG (B_RS + 32)             // 8:  ints [0] bad, [1] ready
#define B_LINV (B_FLAG + 8)            // 256 L_kk^-1 image of the current column
#define B_LINVT (B_LINV + 256)         // 256 L_kk^-T image
#define B_RED (B_LINVT + 256)          // 8192 predict reduction buffer
#define B_TOTAL (B_RED + 8192)         // 20072 doubles = 156.8 KB

__device__ static __forceinline__ d4 bg_mfma4_neg(d4 a, d4 b, d4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 1);   // blgp = 1: NEG(A)
    return acc;
}

__global__ __launch_bounds__(BG_THREADS, 2) void dense_big_kernel(BigParams g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* T = lds + B_EXP;
    double* px0 = lds + B_PX0;
    double* px1 = lds + B_PX1;
    double* zv = lds + B_ZV;
    double* wv = lds + B_WV;
    double* av = lds + B_AV;
    double* rsbuf = lds + B_RS;
    int* flag = reinterpret_cast<int*>(lds + B_FLAG);
    int* ready = flag + 1;
    double* LinvC = lds + B_LINV;
    double* LinvTC = lds + B_LINVT;
    double* red = lds + B_RED;
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
    const unsigned ready_addr = lds0 + (unsigned)(B_FLAG * 8 + 4);

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny), m = A.m, ntw = g.ntw;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    double* Lt = g.ws + (size_t)blockIdx.x * g.slot;              // tiles (i, j): Lt + (i * ntw + j) * 256, i = 0 .. ntw
    double* LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;       // L_kk^-T images, k = 0 .. ntw-1

    gpc_exp_table_init(T);

    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
        const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
        double* fs = A.f_star + (size_t)patch * ny * m;
        __syncthreads();   // previous patch fully done with LDS
        if (n <= 0 || n > MF_TS * ntw) {
            for (int p = tid; p < m * ny; p += BG_THREADS) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            continue;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);
        for (int i = tid; i < BG_NPAD; i += BG_THREADS) {
            const bool live = i < n;
            px0[i] = live ? A.x0[o + i] : 0.0;
            px1[i] = live ? A.x1[o + i] : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c * BG_NPAD + i] = 0.0;
        }
        if (tid == 0) {
            flag[0] = 0;
            flag[1] = -1;
        }
        __syncthreads();
        bool timed_out = false;
        bool bad = false;

        // ---- tiled left-looking Cholesky; tile row nt carries the right-hand sides ----
        for (int k = 0; k < nt; ++k) {
            const int rows_w = (nt - k - wave + BG_WAVES) / BG_WAVES;     // rows k + wave + 8 t <= nt owned by this wave
            for (int p0 = 0; p0 < rows_w || (p0 == 0); p0 += BG_RMAX) {
                const int np = min(BG_RMAX, rows_w - p0);                 // rows in this pass (may be <= 0 for idle waves)
                d4 acc[BG_RMAX];
                // A_rk straight into the accumulators (transposed storage: lane l, register q = A[16 r + (l&15)][16 k + (l>>4) + 4 q])
#pragma unroll
                for (int t = 0; t < BG_RMAX; ++t) {
                    acc[t] = d4{0.0, 0.0, 0.0, 0.0};
                    if (t < np) {
                        const int r = k + wave + BG_WAVES * (p0 + t);
                        if (r < nt) {
                            const int pi = MF_TS * r + lr;
                            const double xi0 = px0[pi], xi1 = px1[pi];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int pj = MF_TS * k + lg + 4 * q;
                                double v = gpc_rbf_neg(sf, cexp, xi0, xi1, px0[pj], px1[pj], T);
                                if (pi == pj) {
                                    v += noise;                              // covariance_matrix(..., training)  :59-61
                                    if (A.prm.ref_double_noise) v += noise;  // C.diagonal() += sigman_sq        :21
                                }
                                if (pi >= n || pj >= n) v = (pi == pj) ? 1.0 : 0.0;     // identity padding
                                acc[t][q] = v;
                            }
                        } else {
                            // right-hand sides: row c = channel, columns = the points of tile column k
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int pj = MF_TS * k + lg + 4 * q;
                                acc[t][q] = (lr < ny && pj < n) ? A.y[(size_t)lr * A.n_total + o + pj] : 0.0;
                            }
                        }
                    }
                }
                // T_rk -= sum_j L_rj L_kj^T; the operand L_kj is loaded once per j for all rows of the pass, j + 1 is prefetched
                if (np > 0 && k > 0) {
                    d4 a_cur = mf_img_load(Lt + ((size_t)k * ntw) * MF_IMG, lane), a_nxt = a_cur;
                    d4 b_cur[BG_RMAX], b_nxt[BG_RMAX];
#pragma unroll
                    for (int t = 0; t < BG_RMAX; ++t) {
                        b_cur[t] = a_cur;
                        if (t < np) b_cur[t] = mf_img_load(Lt + ((size_t)(k + wave + BG_WAVES * (p0 + t)) * ntw) * MF_IMG, lane);
                        b_nxt[t] = b_cur[t];
                    }
                    for (int j = 0; j < k; ++j) {
                        if (j + 1 < k) {
                            a_nxt = mf_img_load(Lt + ((size_t)k * ntw + j + 1) * MF_IMG, lane);
#pragma unroll
                            for (int t = 0; t < BG_RMAX; ++t)
                                if (t < np) b_nxt[t] = mf_img_load(Lt + ((size_t)(k + wave + BG_WAVES * (p0 + t)) * ntw + j + 1) * MF_IMG, lane);
                        }
#pragma unroll
                        for (int t = 0; t < BG_RMAX; ++t)
                            if (t < np) acc[t] = bg_mfma4_neg(a_cur, b_cur[t], acc[t]);
                        a_cur = a_nxt;
#pragma unroll
                        for (int t = 0; t < BG_RMAX; ++t) b_cur[t] = b_nxt[t];
                    }
                }
                // the diagonal tile is row 0 of wave 0's first pass: factor it at once
                if (wave == 0 && p0 == 0) {
                    const bool ok = mf_diag_factor(acc[0], rsbuf, LinvC, LinvTC, g.pivot_tol);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    mf_img_store(LinvTg + (size_t)k * MF_IMG, lane, mf_img_load(LinvTC, lane));    // kept for the backward solve
                    if (!ok && lane == 0) flag[0] = 1;
                    mf_publish(ready, k);
                }
                // L_rk = T_rk L_kk^-T for the other rows
                timed_out |= !mf_wait_ge(ready_addr, k);
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                if (np > 0) {
                    const d4 lv = mf_img_load(LinvC, lane);
#pragma unroll
                    for (int t = 0; t < BG_RMAX; ++t) {
                        if (t < np && !(wave == 0 && p0 == 0 && t == 0)) {
                            const int r = k + wave + BG_WAVES * (p0 + t);
                            const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                            const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], acc[t][0], z4, 0, 0, 0);
                            const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], acc[t][1], z4, 0, 0, 0);
                            const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], acc[t][2], z4, 0, 0, 0);
                            const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], acc[t][3], z4, 0, 0, 0);
                            mf_img_store(Lt + ((size_t)r * ntw + k) * MF_IMG, lane, (D0 + D1) + (D2 + D3));
                        }
                    }
                }
            }
            __syncthreads();   // column k is in the workspace; L_kk^-1 image may be overwritten
            bad = flag[0] != 0;
            if (bad) break;
        }
        if (bad) {
            __syncthreads();
            for (int p = tid; p < m * ny; p += BG_THREADS) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = tid; i < n * ny; i += BG_THREADS)
                    A.alpha_out[(size_t)(i / n) * A.n_total + o + (i % n)] = __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            continue;
        }

        // z_k^T = tile (nt, k): lane l, slot s = z_(l&15)[16 k + (l>>4) + 4 s]
        for (int k = wave; k < nt; k += BG_WAVES) {
            const d4 zt = mf_img_load(Lt + ((size_t)nt * ntw + k) * MF_IMG, lane);
            if (lr < ny) {
#pragma unroll
                for (int s = 0; s < 4; ++s) zv[lr * BG_NPAD + MF_TS * k + lg + 4 * s] = zt[s];
            }
        }
        __syncthreads();

        // ---- backward solve L^T alpha = z, tile columns from the last to the first ----
        for (int k = nt - 1; k >= 0; --k) {
            d4 pa[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) pa[c] = d4{0.0, 0.0, 0.0, 0.0};
            bool any = false;
            for (int i = k + 1 + wave; i < nt; i += BG_WAVES) {
                const d4 li = mf_img_load(Lt + ((size_t)i * ntw + k) * MF_IMG, lane);   // L_ik[l&15][(l>>4) + 4 s]
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) pa[c] += li * av[c * BG_NPAD + MF_TS * i + lr];
                any = true;
            }
            if (any) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (c < ny) {
                        const double tot = mf_row_reduce4(pa[c], lr);     // lanes lr = 0, 4, 8, 12 hold components 0..3
                        if ((lr & 3) == 0) atomicAdd(wv + c * BG_NPAD + MF_TS * k + lg + 4 * (lr >> 2), tot);
                    }
                }
            }
            __syncthreads();
            if (wave == 0) {
                const d4 lt = mf_img_load(LinvTg + (size_t)k * MF_IMG, lane);
                d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                if (lr < ny) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int q = lr * BG_NPAD + MF_TS * k + lg + 4 * q4;
                        ub[q4] = zv[q] - wv[q];
                    }
                }
                const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[0], ub[0], z4, 0, 0, 0);
                const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[1], ub[1], z4, 0, 0, 0);
                const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[2], ub[2], z4, 0, 0, 0);
                const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[3], ub[3], z4, 0, 0, 0);
                const d4 al = (D0 + D1) + (D2 + D3);   // lanes lr = n < ny: alpha_n[16 k + (l>>4) + 4 r]
                if (lr < ny) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[lr * BG_NPAD + MF_TS * k + lg + 4 * r] = al[r];
                }
            }
            __syncthreads();
        }
        if (A.alpha_out)
            for (int i = tid; i < n; i += BG_THREADS)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) A.alpha_out[(size_t)c * A.n_total + o + i] = av[c * BG_NPAD + i];

        // ---- predictive mean ----
        if (A.xs0 == nullptr && A.grid_sz <= 32) {
            // separable grid: f[py][px] = sum_i Ey[py][i] * (sf alpha_i Ex[px][i]); 32-point chunks dealt to the waves
            const int sz = A.grid_sz;
            const double res = A.grid_res;
            for (int c = 0; c < ny; ++c) {
                d4 P[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) P[mt][nl] = d4{0.0, 0.0, 0.0, 0.0};
                for (int ibase = 32 * wave; ibase < n; ibase += 32 * BG_WAVES) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ibase + 4 * s + lg;
                        const double al = sf * av[c * BG_NPAD + i];     // av is zero beyond n (identity padding solves to 0)
                        double ea[2], eb[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int pq = 16 * h + lr;
                            const double gq = res * (((double)pq + 0.5) / (double)sz - 0.5);
                            const bool on = (pq < sz) && (i < n);
                            const double dy = gq - px1[i], dx = gq - px0[i];
                            ea[h] = on ? gpc_exp_neg(cexp * (dy * dy), T) : 0.0;   // Ey[py = pq][i]
                            eb[h] = on ? gpc_exp_neg(cexp * (dx * dx), T) * al : 0.0;   // Ex[px = pq][i] * sf alpha_i
                        }
#pragma unroll
                        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                P[mt][nl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[mt], eb[nl], P[mt][nl], 0, 0, 0);
                    }
                }
                __syncthreads();   // previous channel's reduction finished reading `red`
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl)
                        *reinterpret_cast<d4*>(red + ((wave * 4 + mt * 2 + nl) * 256) + lane * 4) = P[mt][nl];
                __syncthreads();
                for (int oo = tid; oo < 1024; oo += BG_THREADS) {
                    const int tile = oo >> 8, e = oo & 255, l2 = e >> 2, r = e & 3;
                    const int py = 16 * (tile >> 1) + (l2 >> 4) + 4 * r, pxx = 16 * (tile & 1) + (l2 & 15);
                    if (py < sz && pxx < sz) {
                        double s_ = 0.0;
#pragma unroll
                        for (int w = 0; w < BG_WAVES; ++w) s_ += red[(w * 4 + tile) * 256 + e];
                        fs[(size_t)c * m + py * sz + pxx] = s_;
                    }
                }
            }
        } else {
            // point-wise X* (or a grid wider than 32): one thread per prediction point
            for (int p = tid; p < m; p += BG_THREADS) {
                double q0, q1;
                if (A.xs0) {
                    q0 = A.xs0[p];
                    q1 = A.xs1[p];
                } else {
                    const int gx = p % A.grid_sz, gy = p / A.grid_sz;
                    q0 = A.grid_res * (((double)gx + 0.5) / (double)A.grid_sz - 0.5);
                    q1 = A.grid_res * (((double)gy + 0.5) / (double)A.grid_sz - 0.5);
                }
                double s_[3] = {0.0, 0.0, 0.0};
                for (int i = 0; i < n; ++i) {
                    const double kk = gpc_rbf_neg(sf, cexp, px0[i], px1[i], q0, q1, T);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ny) s_[c] += kk * av[c * BG_NPAD + i];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) fs[(size_t)c * m + p] = s_[c];
            }
        }
        if (timed_out && lane == 0) flag[0] = 2;
        __syncthreads();
        if (tid == 0 && A.status) A.status[patch] = flag[0] ? GPC_STATUS_NAN : GPC_STATUS_OK;
    }
}

bool dense_big_supported(const DenseArgs& a)
{
    return a.n_max > 256 && a.n_max <= BG_NPAD && a.v_star == nullptr && (a.ny == 1 || a.ny == 3);
}

static size_t big_slot_doubles(int ntw) { return ((size_t)(ntw + 1) * ntw + ntw) * MF_IMG; }

size_t dense_big_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out)
{
    const int ntw = (a.n_max + MF_TS - 1) / MF_TS;
    const int grid = a.P < ctx->num_cus ? a.P : ctx->num_cus;     // 157 KB of LDS: one workgroup per CU
    if (grid_out) *grid_out = grid;
    return sizeof(double) * big_slot_doubles(ntw) * (size_t)grid;
}

int dense_big_launch(gpc_ctx* ctx, const DenseArgs& a, int grid)
{
    BigParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.ws = static_cast<double*>(ctx->ws);
    g.ntw = (a.n_max + MF_TS - 1) / MF_TS;
    g.slot = big_slot_doubles(g.ntw);
    const size_t lds = sizeof(double) * (size_t)B_TOTAL;
    static bool attr_set = false;
    if (!attr_set) {
        GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(dense_big_kernel, dim3(grid), dim3(BG_THREADS), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    ctx->last_dense_kernel = "dense_mfma_big";
    return GPC_OK;
}
