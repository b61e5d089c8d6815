// dense_mfma_big.hip -- batched dense GP for 256 < n <= 1024 (BASELINE config 3: n = 512, config 5: n = 1024) on gfx950.
//
// Same computation as dense_mfma.hip / dense_generic.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45).  At these sizes the lower triangle of K (1 MB at n = 512, 4 MB at
// n = 1024) exceeds the register file of a CU, so the factor lives in a per-workgroup slot of a global workspace as
// 16 x 16 tiles in MFMA *operand image* layout (mfma_tile.h; 2 KB each, read back with two coalesced 16-byte loads per
// lane) and the factorisation is a tiled LEFT-looking Cholesky:
//
//   columns k, k+1:  T_rc = A_rc - sum_{j<k} L_rj L_cj^T   (4 v_mfma_f64_16x16x4_f64 per (r, c, j)),   r = k .. nt
//              L_kk^-1 by the in-place Gauss-Jordan on the MFMA pipe (mf_diag_factor),   L_rk = T_rk L_kk^-T (4 MFMAs)
//
// * K itself is never stored: A_rk is evaluated (RBF + noise diagonal) straight into the accumulator registers when
//   column k starts.  HBM/L2 see the factor only: one write per tile, nt/3 reads on average.
// * TWO tile columns per step: every fetched tile L_rj feeds the accumulators of both columns (the kernel streams the
//   factor from HBM / Infinity Cache: 4 TB/s measured, it was 5.8 TB/s and 25 % slower with one column per step).  Tile
//   rows are dealt round-robin to the waves; a wave keeps up to BG_RMAX rows (x 2 columns) of accumulators, loads the
//   shared operands L_kj, L_(k+1)j once per j for all of them, and keeps the loads of the next two j in flight.
// * The right-hand sides ride along as one more tile row (row nt holds y^T padded to 16 channels): its TRSM result is
//   z^T, so the forward solve needs no code of its own.
// * Wave 0 owns the 2 x 2 diagonal block of the step (update, factor, TRSM, update, factor) while the other waves stream
//   their row updates; they poll an LDS word for each L^-1 (bounded asm poll, mfma_tile.h).  One workgroup barrier per step.
// * The kernel is a template over waves per workgroup and padded size: <8, 1024> is the production shape (one workgroup
//   per CU); <4, 256> runs two workgroups per CU and is kept as a cross-check of dense_mfma.hip (GPC_FORCE_BIG=1).
// * Backward solve: alpha_k = L_kk^-T (z_k - sum_{i>k} L_ik^T alpha_i); the tile products contract over the ROW index,
//   which the image layout cannot feed to an MFMA, so they run on the VALU with the DPP row reduction of mfma_tile.h
//   (O(n^2) work against the O(n^3) of the factorisation).  Predictive mean: the separable-grid MFMA form of
//   dense_mfma.hip, looped over 32-point chunks.
#include <cstdio>
#include <cstdlib>

#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"


struct BigParams {
    DenseArgs a;
    double c_exp;
    double pivot_tol;
    double* ws;
    size_t slot;   // doubles per workgroup slot
    int ntw;       // tile columns of a slot = ceil(n_max / 16)
    unsigned long long* stamps;   // diagnostic (GPC_BIG_STAMPS=1): [phase][wave] cycle sums over all patches, else nullptr
    int export_factor;            // predictive variance (dense_variance.hip): the factor of EVERY patch stays in the workspace (slot =
                                  // patch, not workgroup) together with the L_kk^-1 images
    // IRLS instantiation only (BASELINE config 5, gpc_dense_irls_fit_predict): the Newton loop around the factorisation
    int irls_model;               // GPC_NOISE_PROBIT_REF / GPC_NOISE_PROBIT_STD
    int irls_max_iter;
    double irls_tol, irls_f_init;
    int32_t* irls_iters;          // [P] solves performed, or nullptr
    double* irls_fhat;            // [n_total] latent mode at the training points, or nullptr
};
#define BG_NPH 6
#define BG_STAMP(ph)                                                                                                 \
    do {                                                                                                             \
        if (g.stamps) {                                                                                              \
            const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                                          \
            if (lane == 0) atomicAdd(g.stamps + (ph) * 8 + wave, t_now_ - t_prev_);                                  \
            t_prev_ = t_now_;                                                                                        \
        }                                                                                                            \
    } while (0)

// LDS carve (doubles), for NPAD padded points and WAVES waves per workgroup:
//   exp table 64 | x0, x1 2 NPAD | z, w, alpha 9 NPAD | rsqrt row 32 | flags 8 | 2 L^-1 images 512 | L_(k+1)k image 256 |
//   predict accumulation buffer 4 x 256 (ds_add_f64 targets; during the factorisation: hand-over of the diagonal-block tiles)
__host__ __device__ constexpr int bg_lds_doubles(int npad, int waves) { return 64 + 11 * npad + 32 + 8 + 512 + 256 + 1024 + 0 * waves; }

__device__ static __forceinline__ d4 bg_mfma4_neg(d4 a, d4 b, d4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 1);   // blgp = 1: NEG(A)
    return acc;
}

// BG_WAVES waves per workgroup and BG_NPAD padded points: <8, 1024> (one workgroup per CU) for 256 < n <= 1024.
//
// BG_IRLS: the same factorisation inside a Newton / IRLS loop for a non-Gaussian likelihood (probit_noise,
// /root/reference/src/probit_noise.cpp:11-31; BASELINE config 5).  Every Newton step of the Laplace mode search is one dense
// GP regression fit with per-point noise d_i = 1 / W_i and working targets t_i = f_i + g_i d_i (W = -dx2_ln, g = dx_ln of the
// functor at sigma_x = 0):  a = (K + diag d)^-1 t,  f_new = K a = t - d o a  -- so the loop re-runs the tiled Cholesky with the
// diagonal and the right-hand-side row taken from LDS vectors instead of the scalar noise and y, and needs no mat-vec with
// K at all.  What is factored is the symmetrically scaled matrix B = I + W^1/2 K W^1/2 (Rasmussen & Williams eq. 3.26:
// eigenvalues >= 1 however far the weights spread; the tile TRSMs here multiply by explicit 16 x 16 inverses, which is only
// as accurate as the tiles are well conditioned): u = B^-1 W^1/2 t, a = W^1/2 u -- the same Newton iterate.  Definition and stopping rule: oracle/gpc_oracle.c (orc_dense_irls_fit).  ny == 1; the unused colour planes of
// the solve vectors hold f, d, t and the labels.
template <int BG_WAVES, int BG_NPAD, int BG_RMAX, int BG_OCC, bool BG_IRLS = false>
__global__ __launch_bounds__(BG_WAVES * 64, BG_OCC) void dense_big_kernel(BigParams g)
{
    static_assert(BG_RMAX == 1 || BG_RMAX == 2, "one copy of the update loop per possible row count");
    constexpr int BG_THREADS = BG_WAVES * 64;
    constexpr int B_PX0 = 64, B_PX1 = B_PX0 + BG_NPAD, B_ZV = B_PX1 + BG_NPAD, B_WV = B_ZV + 3 * BG_NPAD, B_AV = B_WV + 3 * BG_NPAD,
                  B_RS = B_AV + 3 * BG_NPAD, B_FLAG = B_RS + 32, B_LINV = B_FLAG + 8, B_L10 = B_LINV + 512, B_RED = B_L10 + 256;
    static_assert(B_RED + 1024 == bg_lds_doubles(BG_NPAD, BG_WAVES), "LDS carve");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* T = lds;
    double* px0 = lds + B_PX0;
    double* px1 = lds + B_PX1;
    double* zv = lds + B_ZV;
    double* wv = lds + B_WV;
    double* av = lds + B_AV;
    double* rsbuf = lds + B_RS;
    int* flag = reinterpret_cast<int*>(lds + B_FLAG);   // ints [0] bad, [1] ready
    int* ready = flag + 1;
    double* LinvC = lds + B_LINV;      // [2][256]
    double* L10 = lds + B_L10;
    double* red = lds + B_RED;
    double* Hand = red;                // 3 x 256: hand-over of the diagonal-block tiles to wave 0 (the reduction buffer is idle then)
    // IRLS (ny == 1): planes 1, 2 of the solve vectors are free
    [[maybe_unused]] double* fv = zv + BG_NPAD;         // latent f
    [[maybe_unused]] double* dv = zv + 2 * BG_NPAD;     // W^-1/2
    [[maybe_unused]] double* sv = av + BG_NPAD;         // W^1/2
    [[maybe_unused]] double* tv = wv + BG_NPAD;         // working targets
    [[maybe_unused]] double* yl = wv + 2 * BG_NPAD;     // labels
    [[maybe_unused]] unsigned long long* delta_bits = reinterpret_cast<unsigned long long*>(lds + B_FLAG + 4);   // max |f_new - f| of the step, as bits
    int* h0 = flag + 2;                // k + 1 once tile (k, k) of the step is handed over
    int* h1 = flag + 3;                // k + 1 once tiles (k+1, k), (k+1, k+1) are handed over
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
    const unsigned ready_addr = lds0 + (unsigned)(B_FLAG * 8 + 4);     // highest tile column whose L_kk^-1 is published
    const unsigned h0_addr = lds0 + (unsigned)(B_FLAG * 8 + 8), h1_addr = lds0 + (unsigned)(B_FLAG * 8 + 12);

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny), m = A.m, ntw = g.ntw;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    double* Lt = g.ws + (size_t)blockIdx.x * g.slot;              // tiles (i, j): Lt + (i * ntw + j) * 256, i = 0 .. ntw
    double* LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;       // L_kk^-T images, k = 0 .. ntw-1
    double* LinvG = LinvTg + (size_t)ntw * MF_IMG;                // L_kk^-1 images (written when the factor is exported)

    gpc_exp_table_init(T);

    const int n_patches = A.sel ? __builtin_amdgcn_readfirstlane(A.sel_count[0]) : A.P;   // size-class dispatch: sel[0 .. count)
    for (int pk = blockIdx.x; pk < n_patches; pk += gridDim.x) {
        const int patch = A.sel ? __builtin_amdgcn_readfirstlane(A.sel[pk]) : pk;
        if (g.export_factor) {
            Lt = g.ws + (size_t)patch * g.slot;
            LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;
            LinvG = LinvTg + (size_t)ntw * MF_IMG;
        }
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
        unsigned long long t_prev_ = g.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned long long* ext_bits = reinterpret_cast<unsigned long long*>(lds + B_FLAG + 3);   // max-norm extent, as bits
        if (tid == 0) {
            flag[0] = 0;
            flag[1] = -1;
            flag[2] = 0;
            flag[3] = 0;
            *ext_bits = 0ull;
        }
        __syncthreads();
        // extent of the patch around its first point (max-norm): bounds every kernel argument of this patch, see below
        const double xo0 = A.x0[o], xo1 = A.x1[o];
        double dev = 0.0;
        for (int i = tid; i < BG_NPAD; i += BG_THREADS) {
            const bool live = i < n;
            const double q0 = live ? A.x0[o + i] : xo0, q1 = live ? A.x1[o + i] : xo1;
            dev = __builtin_fmax(dev, __builtin_fmax(__builtin_fabs(q0 - xo0), __builtin_fabs(q1 - xo1)));
            px0[i] = live ? q0 : 0.0;
            px1[i] = live ? q1 : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) { wv[c * BG_NPAD + i] = 0.0; av[c * BG_NPAD + i] = 0.0; }   // av: no stale LDS beyond the solved rows
        }
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) dev = __builtin_fmax(dev, __shfl_xor(dev, o_, 64));
        if (lane == 0) atomicMax(ext_bits, (unsigned long long)__double_as_longlong(dev));   // non-negative doubles order like their bits
        __syncthreads();
        // Small-argument regime (see dense_mfma.hip): |c| d^2 <= 2^-5 for every Gram argument / every separable grid factor
        // -> the degree-7 polynomial instead of the table-driven exponential.  Wave-uniform; false for NaN / inf extents.
        bool small_gram, small_grid;
        {
            const double r = __longlong_as_double((long long)*ext_bits);
            const double bq = 0.5 * A.grid_res + __builtin_fmax(__builtin_fabs(xo0), __builtin_fabs(xo1)) + r;
            small_gram = __builtin_amdgcn_readfirstlane((int)(-cexp * (8.0 * r * r) <= GPC_EXP_SMALL_MAX)) != 0;
            small_grid = __builtin_amdgcn_readfirstlane((int)(-cexp * (bq * bq) <= GPC_EXP_SMALL_MAX)) != 0;
        }
        bool timed_out = false;
        bool bad = false;
        BG_STAMP(0);

        // ---- tiled left-looking Cholesky, two tile columns (k, k+1) per step; tile row nt carries the right-hand sides ----
        // Two columns per step halve the dominant HBM stream: every tile L_rj fetched for the update of row r feeds the
        // accumulators of both columns (8 MFMAs per 2 KB instead of 4; measured 5.8 TB/s with one column per step).
        // Wave 0 owns the 2 x 2 diagonal block (tiles (k,k), (k+1,k), (k+1,k+1)): update -> factor (k,k) -> TRSM (k+1,k) ->
        // update and factor (k+1,k+1), publishing each L^-1 as it appears.  The rows r >= k+2 (and the right-hand-side row
        // nt) are dealt to the waves 1, 2, .., 7, 0, 1, ..: update both accumulators, L_rk = TRSM(acc0), acc1 -= L_rk L_(k+1)k^T,
        // L_r(k+1) = TRSM(acc1).
#define BG_INIT_TILE(dst, r_, kc_)                                                                                   \
    do {                                                                                                             \
        if ((r_) < nt) {                                                                                             \
            const int pi_ = MF_TS * (r_) + lr;                                                                       \
            const double xi0_ = px0[pi_], xi1_ = px1[pi_];                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
                const int pj_ = MF_TS * (kc_) + lg + 4 * q_;                                                         \
                double v_ = small_gram ? gpc_rbf_small(sf, cexp, xi0_, xi1_, px0[pj_], px1[pj_])                      \
                                       : gpc_rbf_neg(sf, cexp, xi0_, xi1_, px0[pj_], px1[pj_], T);                   \
                if constexpr (BG_IRLS) v_ = (v_ * sv[pi_]) * sv[pj_];                                                \
                if (pi_ == pj_) {                                                                                    \
                    if constexpr (BG_IRLS) {                                                                         \
                        v_ += 1.0;                                /* B = I + W^1/2 K W^1/2 */                        \
                    } else {                                                                                         \
                        v_ += noise;                              /* covariance_matrix(..., training)  :59-61 */     \
                        if (A.prm.ref_double_noise) v_ += noise;  /* C.diagonal() += sigman_sq        :21 */         \
                    }                                                                                                \
                }                                                                                                    \
                if (pi_ >= n || pj_ >= n) v_ = (pi_ == pj_) ? 1.0 : 0.0;     /* identity padding */                  \
                (dst)[q_] = v_;                                                                                      \
            }                                                                                                        \
        } else {   /* right-hand sides: row c = channel, columns = the points of tile column kc_ */                  \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
                const int pj_ = MF_TS * (kc_) + lg + 4 * q_;                                                         \
                if constexpr (BG_IRLS) (dst)[q_] = (lr == 0 && pj_ < n) ? sv[pj_] * tv[pj_] : 0.0;                   \
                else (dst)[q_] = (lr < ny && pj_ < n) ? A.y[(size_t)lr * A.n_total + o + pj_] : 0.0;                 \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define BG_TRSM(lv_, src_)                                                                                           \
    ((__builtin_amdgcn_mfma_f64_16x16x4f64((lv_)[0], (src_)[0], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0) +                   \
      __builtin_amdgcn_mfma_f64_16x16x4f64((lv_)[1], (src_)[1], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0)) +                  \
     (__builtin_amdgcn_mfma_f64_16x16x4f64((lv_)[2], (src_)[2], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0) +                   \
      __builtin_amdgcn_mfma_f64_16x16x4f64((lv_)[3], (src_)[3], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0)))
        if constexpr (BG_IRLS) {
            for (int i = tid; i < BG_NPAD; i += BG_THREADS) {
                const double lab = (i < n) ? A.y[o + i] : 0.0;
                yl[i] = lab;
                fv[i] = lab * g.irls_f_init;
            }
            if (tid == 0) flag[4] = 0;
            __syncthreads();
        }
        int iter = 0;
        bool nan_w = false;
        [[maybe_unused]] bool converged = true;      // IRLS: false when the step cap ended the loop
        for (;;) {     // Newton / IRLS iterations (BG_IRLS); a single pass otherwise
        if constexpr (BG_IRLS) {
            // weights and working targets from the functor at the current f (sigma_x = 0); reset the step's hand-over words
            int badw = 0;
            for (int i = tid; i < BG_NPAD; i += BG_THREADS) {
                double rs = 0.0, ss = 0.0, tt = 0.0;
                if (i < n) {
                    double q_, r_;
                    gpc_probit_q_r(g.irls_model, noise, yl[i], fv[i], 0.0, &q_, &r_);
                    const double W = -r_;
                    if (!(W > 0.0) || !(W < __builtin_inf()) || q_ != q_) badw = 1;
                    tt = fv[i] + q_ * (1.0 / W);
                    ss = sqrt(W);
                    rs = 1.0 / ss;
                }
                dv[i] = rs;
                sv[i] = ss;
                tv[i] = tt;
                wv[i] = 0.0;
            }
            if (badw) flag[4] = 1;
            if (tid == 0) {
                flag[0] = 0;
                flag[1] = -1;
                flag[2] = 0;
                flag[3] = 0;
                *delta_bits = 0ull;
            }
            __syncthreads();
            nan_w = flag[4] != 0;
            if (nan_w) break;
        }
        for (int k = 0; k < nt; k += 2) {
            const bool has2 = k + 1 < nt;
            const int k1 = has2 ? k + 1 : k;                               // second column of the pair (== k when absent)
            const double* rowk = Lt + ((size_t)k * ntw) * MF_IMG;          // L_kj, j < k
            const double* rowk1 = Lt + ((size_t)k1 * ntw) * MF_IMG;
            // Rows k .. nt are dealt round-robin to the WORKER waves 1 .. W-1 (row k + q -> wave 1 + q % (W-1)).  Rows k and
            // k+1 are the diagonal block: their owners update them like any other row (first pass), then hand the tiles
            // (k,k), (k+1,k), (k+1,k+1) to wave 0 through LDS instead of solving them.  Wave 0 runs nothing but the serial
            // chain factor (k,k) -> TRSM (k+1,k) -> update, factor (k+1,k+1), concurrently with the workers' later passes.
            // (With the diagonal-block updates on wave 0 itself it was the critical path: 3 tile updates per j on one wave.)
            constexpr int NWK = BG_WAVES - 1;
            if (wave == 0) {
                timed_out |= !mf_wait_ge(h0_addr, k + 1);
                const d4 D00 = *reinterpret_cast<const d4*>(Hand + lane * 4);
                bool ok = mf_diag_factor(D00, rsbuf, LinvC, LinvTg + (size_t)k * MF_IMG, g.pivot_tol);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (g.export_factor) mf_img_store(LinvG + (size_t)k * MF_IMG, lane, mf_img_load(LinvC, lane));
                if (!ok && lane == 0) flag[0] = 1;
                if (ok && has2) {
                    mf_publish(ready, k);                                  // L_kk^-1 (the workers start their first TRSM)
                    timed_out |= !mf_wait_ge(h1_addr, k + 1);
                    const d4 D10 = *reinterpret_cast<const d4*>(Hand + 256 + lane * 4);
                    d4 D11 = *reinterpret_cast<const d4*>(Hand + 512 + lane * 4);
                    const d4 lv = mf_img_load(LinvC, lane);
                    const d4 l10 = BG_TRSM(lv, D10);                       // operand image of L_(k+1)k
                    mf_img_store(Lt + ((size_t)(k + 1) * ntw + k) * MF_IMG, lane, l10);
                    mf_img_store(L10, lane, l10);
                    D11 = bg_mfma4_neg(l10, l10, D11);
                    ok = mf_diag_factor(D11, rsbuf, LinvC + 256, LinvTg + (size_t)(k + 1) * MF_IMG, g.pivot_tol);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (g.export_factor) mf_img_store(LinvG + (size_t)(k + 1) * MF_IMG, lane, mf_img_load(LinvC + 256, lane));
                    if (!ok && lane == 0) flag[0] = 1;
                }
                mf_publish(ready, k1);
            } else {
                const int q0 = wave - 1;                                   // first row of this worker: k + q0
                const int rows_tot = nt - k + 1;                           // rows k .. nt
                const int rows_w = (rows_tot - 1 - q0 >= 0) ? (rows_tot - 1 - q0) / NWK + 1 : 0;
                bool stop = false;
                for (int p0 = 0; (p0 < rows_w || p0 == 0) && !stop; p0 += BG_RMAX) {
                    const int np = min(BG_RMAX, rows_w - p0);
                    d4 acc0[BG_RMAX], acc1[BG_RMAX];
                    int rr[BG_RMAX];
#pragma unroll
                    for (int t = 0; t < BG_RMAX; ++t) {
                        rr[t] = k + q0 + NWK * (p0 + t);
                        acc0[t] = d4{0.0, 0.0, 0.0, 0.0};
                        acc1[t] = d4{0.0, 0.0, 0.0, 0.0};
                        if (t < np) {
                            BG_INIT_TILE(acc0[t], rr[t], k);
                            if (has2 && rr[t] != k) BG_INIT_TILE(acc1[t], rr[t], k + 1);     // (k, k+1) is above the diagonal
                        }
                    }
                    if (np > 0 && k > 0) {
                        // three operand stages rotate through the loop (unrolled by 3): while stage s feeds the MFMAs, the
                        // loads of the next two j are in flight -- the factor tiles come from HBM / Infinity Cache
                        d4 sa0[3], sa1[3], sb[3][BG_RMAX];
                        // NPC (rows in the pass) is a compile-time constant inside each copy of the loop and the loads are
                        // UNCONDITIONAL (index clamped to k-1, a redundant re-read at the tail): with loads under runtime
                        // conditions hipcc cannot count the outstanding ones and falls back to s_waitcnt vmcnt(0) in every
                        // iteration, i.e. no prefetch at all
#define BG_LOAD_STAGE(st, jj, NPC)                                                                                   \
    do {                                                                                                             \
        sa0[st] = mf_img_load(rowk + (size_t)(jj) * MF_IMG, lane);                                                   \
        sa1[st] = mf_img_load(rowk1 + (size_t)(jj) * MF_IMG, lane);                                                  \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                              \
            sb[st][t] = mf_img_load(Lt + ((size_t)rr[t] * ntw + (jj)) * MF_IMG, lane);                               \
    } while (0)
#define BG_USE_STAGE(st, NPC)                                                                                        \
    do {                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t) {                                                            \
            acc0[t] = bg_mfma4_neg(sa0[st], sb[st][t], acc0[t]);                                                     \
            if (has2 && rr[t] != k) acc1[t] = bg_mfma4_neg(sa1[st], sb[st][t], acc1[t]);                             \
        }                                                                                                            \
    } while (0)
#define BG_UPDATE_LOOP(NPC)                                                                                          \
    do {                                                                                                             \
        const int kl = k - 1;                                                                                        \
        BG_LOAD_STAGE(0, 0, NPC);                                                                                    \
        BG_LOAD_STAGE(1, min(1, kl), NPC);                                                                           \
        for (int j = 0; j < k; j += 3) {                                                                             \
            BG_LOAD_STAGE(2, min(j + 2, kl), NPC);                                                                   \
            BG_USE_STAGE(0, NPC);                                                                                    \
            BG_LOAD_STAGE(0, min(j + 3, kl), NPC);                                                                   \
            if (j + 1 < k) BG_USE_STAGE(1, NPC);                                                                     \
            BG_LOAD_STAGE(1, min(j + 4, kl), NPC);                                                                   \
            if (j + 2 < k) BG_USE_STAGE(2, NPC);                                                                     \
        }                                                                                                            \
    } while (0)
#pragma unroll
                        for (int st = 0; st < 3; ++st) {
                            sa0[st] = d4{0.0, 0.0, 0.0, 0.0};
                            sa1[st] = sa0[st];
#pragma unroll
                            for (int t = 0; t < BG_RMAX; ++t) sb[st][t] = sa0[st];
                        }
                        if constexpr (BG_RMAX == 2) {
                            if (np == 2) BG_UPDATE_LOOP(2);
                            else BG_UPDATE_LOOP(1);
                        } else {
                            BG_UPDATE_LOOP(1);
                        }
                    }
                    // diagonal-block rows go to wave 0 (first pass only: q = 0 -> wave 1, q = 1 -> wave 2, both t = 0)
                    if (p0 == 0 && np > 0 && rr[0] == k) {
                        *reinterpret_cast<d4*>(Hand + lane * 4) = acc0[0];
                        mf_publish(h0, k + 1);
                    }
                    if (p0 == 0 && np > 0 && has2 && rr[0] == k + 1) {
                        *reinterpret_cast<d4*>(Hand + 256 + lane * 4) = acc0[0];
                        *reinterpret_cast<d4*>(Hand + 512 + lane * 4) = acc1[0];
                        mf_publish(h1, k + 1);
                    }
                    // L_rk = T_rk L_kk^-T;  T_r(k+1) -= L_rk L_(k+1)k^T;  L_r(k+1) = T_r(k+1) L_(k+1)(k+1)^-T
                    timed_out |= !mf_wait_ge(ready_addr, k);
                    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { stop = true; continue; }
                    if (np > 0) {
                        const d4 lv0 = mf_img_load(LinvC, lane);
#pragma unroll
                        for (int t = 0; t < BG_RMAX; ++t) {
                            if (t < np && rr[t] > k1) {
                                acc0[t] = BG_TRSM(lv0, acc0[t]);
                                mf_img_store(Lt + ((size_t)rr[t] * ntw + k) * MF_IMG, lane, acc0[t]);
                            }
                        }
                    }
                    if (has2) {
                        timed_out |= !mf_wait_ge(ready_addr, k + 1);
                        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { stop = true; continue; }
                        if (np > 0) {
                            const d4 lv1 = mf_img_load(LinvC + 256, lane);
                            const d4 a10 = mf_img_load(L10, lane);
#pragma unroll
                            for (int t = 0; t < BG_RMAX; ++t) {
                                if (t < np && rr[t] > k1) {
                                    acc1[t] = bg_mfma4_neg(a10, acc0[t], acc1[t]);
                                    mf_img_store(Lt + ((size_t)rr[t] * ntw + k + 1) * MF_IMG, lane, BG_TRSM(lv1, acc1[t]));
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();   // the column pair is in the workspace; the L^-1 images may be overwritten
            bad = flag[0] != 0;
            if (bad) break;
        }
        if (bad) break;
        BG_STAMP(1);
        // z_k^T = tile (nt, k): lane l, slot s = z_(l&15)[16 k + (l>>4) + 4 s]
        for (int k = wave; k < nt; k += BG_WAVES) {
            const d4 zt = mf_img_load(Lt + ((size_t)nt * ntw + k) * MF_IMG, lane);
            if (lr < ny) {
#pragma unroll
                for (int s = 0; s < 4; ++s) zv[lr * BG_NPAD + MF_TS * k + lg + 4 * s] = zt[s];
            }
        }
        __syncthreads();

        BG_STAMP(2);
        // ---- backward solve L^T alpha = z, tile columns from the last to the first ----
        // Column k needs the tiles (i, k), i = k+1+wave+W t, and L_kk^-T: they do not depend on alpha, so the loads of
        // column k-1 are issued before the products of column k (the factor sits in HBM / Infinity Cache, ~2k cycles away).
        {
            constexpr int BT = (BG_NPAD / MF_TS + BG_WAVES - 1) / BG_WAVES;      // tiles per wave and column, at most
            d4 cur[BT], nxt[BT], lt_cur = d4{0.0, 0.0, 0.0, 0.0}, lt_nxt = lt_cur;
#pragma unroll
            for (int t = 0; t < BT; ++t) cur[t] = nxt[t] = d4{0.0, 0.0, 0.0, 0.0};
            if (wave == 0) lt_cur = mf_img_load(LinvTg + (size_t)(nt - 1) * MF_IMG, lane);
            for (int k = nt - 1; k >= 0; --k) {
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < BT; ++t) {
                        const int i = k + wave + BG_WAVES * t;               // rows of column k-1: i >= k
                        if (i < nt) nxt[t] = mf_img_load(Lt + ((size_t)i * ntw + (k - 1)) * MF_IMG, lane);
                    }
                    if (wave == 0) lt_nxt = mf_img_load(LinvTg + (size_t)(k - 1) * MF_IMG, lane);
                }
                d4 pa[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) pa[c] = d4{0.0, 0.0, 0.0, 0.0};
                bool any = false;
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const int i = k + 1 + wave + BG_WAVES * t;
                    if (i < nt) {                                             // cur[t] = L_ik[l&15][(l>>4) + 4 s]
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (c < ny) pa[c] += cur[t] * av[c * BG_NPAD + MF_TS * i + lr];
                        any = true;
                    }
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
                    d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                    if (lr < ny) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const int q = lr * BG_NPAD + MF_TS * k + lg + 4 * q4;
                            ub[q4] = zv[q] - wv[q];
                        }
                    }
                    const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                    const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt_cur[0], ub[0], z4, 0, 0, 0);
                    const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt_cur[1], ub[1], z4, 0, 0, 0);
                    const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt_cur[2], ub[2], z4, 0, 0, 0);
                    const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt_cur[3], ub[3], z4, 0, 0, 0);
                    const d4 al = (D0 + D1) + (D2 + D3);   // lanes lr = n < ny: alpha_n[16 k + (l>>4) + 4 r]
                    if (lr < ny) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) av[lr * BG_NPAD + MF_TS * k + lg + 4 * r] = al[r];
                    }
                }
                __syncthreads();
                // rows of column k-1 held in nxt[t] are i = k + wave + W t; as cur[] of the next iteration they must sit at
                // i = (k-1) + 1 + wave + W t: the same index
#pragma unroll
                for (int t = 0; t < BT; ++t) cur[t] = nxt[t];
                lt_cur = lt_nxt;
            }
        }
        if constexpr (!BG_IRLS) break;
        if constexpr (BG_IRLS) {
            // the solve gave u = B^-1 W^1/2 t:  a = W^1/2 u,  f_new = K a = t - W^-1 a = t - W^-1/2 u;  the step's max |f_new - f|
            // decides (workgroup-uniform, through LDS)
            double dmax = 0.0;
            for (int i = tid; i < n; i += BG_THREADS) {
                const double u_ = av[i];
                av[i] = sv[i] * u_;
                const double fn = tv[i] - dv[i] * u_;
                const double df = __builtin_fabs(fn - fv[i]);
                dmax = (df > dmax || df != df) ? df : dmax;
                fv[i] = fn;
            }
            // non-negative doubles (and NaN, above every finite value) order like their bit patterns
            unsigned long long db = (unsigned long long)__double_as_longlong(dmax);
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) {
                const unsigned long long ob = (unsigned long long)__shfl_xor((long long)db, o_, 64);
                db = ob > db ? ob : db;
            }
            if (lane == 0) atomicMax(delta_bits, db);
            __syncthreads();
            const double delta = __longlong_as_double((long long)*delta_bits);
            ++iter;
            if (delta != delta) { nan_w = true; break; }
            if (delta <= g.irls_tol) break;
            if (iter >= g.irls_max_iter) { converged = false; break; }
            __syncthreads();   // delta_bits is reset by the next prologue
        }
        }   // Newton / IRLS iterations
        if (bad || nan_w) {
            __syncthreads();
            if constexpr (BG_IRLS) {
                if (g.irls_fhat)
                    for (int i = tid; i < n; i += BG_THREADS) g.irls_fhat[o + i] = __builtin_nan("");
                if (tid == 0 && g.irls_iters) g.irls_iters[patch] = iter;
            }
            for (int p = tid; p < m * ny; p += BG_THREADS) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = tid; i < n * ny; i += BG_THREADS)
                    A.alpha_out[(size_t)(i / n) * A.n_total + o + (i % n)] = __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = nan_w ? GPC_STATUS_NAN : GPC_STATUS_NOT_SPD;
            continue;
        }

        if (A.alpha_out)
            for (int i = tid; i < n; i += BG_THREADS)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) A.alpha_out[(size_t)c * A.n_total + o + i] = av[c * BG_NPAD + i];

        if constexpr (BG_IRLS) {
            if (g.irls_fhat)
                for (int i = tid; i < n; i += BG_THREADS) g.irls_fhat[o + i] = fv[i];
            if (tid == 0 && g.irls_iters) g.irls_iters[patch] = iter;
        }

        BG_STAMP(3);
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
                        // av is zero from n to 16 nt (identity padding solves to 0) but never written beyond: stale LDS there
                        // (possibly NaN from the previous kernel on this CU) must not reach the sum -- select, do not multiply
                        const double al = (i < n) ? sf * av[c * BG_NPAD + i] : 0.0;
                        double ea[2], eb[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int pq = 16 * h + lr;
                            const double gq = res * (((double)pq + 0.5) / (double)sz - 0.5);
                            const bool on = (pq < sz) && (i < n);
                            const double dy = gq - px1[i], dx = gq - px0[i];
                            if (small_grid) {
                                ea[h] = on ? gpc_exp_small(cexp * (dy * dy)) : 0.0;         // Ey[py = pq][i]
                                eb[h] = on ? gpc_exp_small(cexp * (dx * dx)) * al : 0.0;    // Ex[px = pq][i] * sf alpha_i
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
                // cross-wave sum of the four 16 x 16 output tiles with ds_add_f64 into one 8 KB buffer
                __syncthreads();   // previous channel's outputs are read; the hand-over scratch of the factorisation is dead
                for (int oo = tid; oo < 1024; oo += BG_THREADS) red[oo] = 0.0;
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(red + (mt * 2 + nl) * 256 + lane * 4 + r, P[mt][nl][r]);
                __syncthreads();
                for (int oo = tid; oo < 1024; oo += BG_THREADS) {
                    const int tile = oo >> 8, e = oo & 255, l2 = e >> 2, r = e & 3;
                    const int py = 16 * (tile >> 1) + (l2 >> 4) + 4 * r, pxx = 16 * (tile & 1) + (l2 & 15);
                    if (py < sz && pxx < sz) fs[(size_t)c * m + py * sz + pxx] = red[oo];
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
        BG_STAMP(4);
        if (timed_out && lane == 0) flag[0] = 2;
        __syncthreads();
        if (tid == 0 && A.status) A.status[patch] = flag[0] ? GPC_STATUS_NAN : (BG_IRLS && !converged) ? GPC_STATUS_NOT_CONVERGED : GPC_STATUS_OK;
    }
}

bool dense_big_supported(const DenseArgs& a)
{
    // (with the variance: point-wise X* only -- the variance entry has no grid form)
    return a.n_max > 256 && a.n_max <= 1024 && (a.v_star == nullptr || a.xs0 != nullptr) && (a.ny == 1 || a.ny == 3);
}

size_t big_slot_doubles(int ntw) { return ((size_t)(ntw + 1) * ntw + 2 * (size_t)ntw) * MF_IMG; }

// <8 waves, 1024 points, 2 rows per pass, 2 waves/SIMD>: 103 KB of LDS, one workgroup per CU: 256 < n <= 1024.
// <4 waves, 256 points, 2 rows per pass, 2 waves/SIMD>: 37 KB of LDS, two workgroups = two patches per CU: the cross-check
// shape for n <= 256 (GPC_FORCE_BIG=1).  (A 1-row, <= 128-VGPR variant with FOUR patches per CU was measured slower, 2.45 M
// against 2.68 M patches/s on C2: each workgroup runs 2.2x longer -- the shape is bound by the factor stream, not by latency.)
static void big_shape(const DenseArgs& a, int* waves, int* npad, int* per_cu)
{
    if (a.n_max <= 256) { *waves = 4; *npad = 256; *per_cu = 2; }
    else { *waves = 8; *npad = 1024; *per_cu = 1; }
}

size_t dense_big_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out)
{
    int waves, npad, per_cu;
    big_shape(a, &waves, &npad, &per_cu);
    const int ntw = (a.n_max + MF_TS - 1) / MF_TS;
    int cap = ctx->num_cus * per_cu;
    if (const char* e = getenv("GPC_BIG_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;   // diagnostic: resident-workgroup experiments
    const int grid = a.P < cap ? a.P : cap;
    if (grid_out) *grid_out = grid;
    // with the variance requested the factor of every patch is kept (one slot per patch) + alpha + the per-wave V scratch of
    // dense_variance_big_kernel
    const size_t slots = a.v_star ? (size_t)a.P : (size_t)grid;
    const size_t extra = a.v_star ? sizeof(double) * ((size_t)a.n_total * a.ny + dense_variance_big_scratch_doubles(ctx, ntw)) : 0;
    return sizeof(double) * big_slot_doubles(ntw) * slots + extra;
}

template <int W, int NP, int RM, int OC, bool IRLS = false>
static int big_launch_t(gpc_ctx* ctx, const BigParams& g, int grid)
{
    const size_t lds = sizeof(double) * (size_t)bg_lds_doubles(NP, W);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_big_kernel<W, NP, RM, OC, IRLS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((dense_big_kernel<W, NP, RM, OC, IRLS>), dim3(grid), dim3(W * 64), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

// BASELINE config 5: the Newton / IRLS loop around the tiled factorisation (any n <= 1024; the 4-wave shape for n <= 256)
int dense_irls_launch(gpc_ctx* ctx, const DenseArgs& a, const IrlsArgs& ir, int grid)
{
    BigParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL;                        // B = I + W^1/2 K W^1/2: every pivot is >= 1 in exact arithmetic
    g.ws = static_cast<double*>(ctx->ws);
    g.ntw = (a.n_max + MF_TS - 1) / MF_TS;
    g.slot = big_slot_doubles(g.ntw);
    g.stamps = nullptr;
    g.export_factor = 0;
    g.irls_model = a.prm.noise_model;
    g.irls_max_iter = ir.max_iter;
    g.irls_tol = ir.tol;
    g.irls_f_init = ir.f_init;
    g.irls_iters = ir.iters;
    g.irls_fhat = ir.fhat;
    int waves, npad, per_cu;
    big_shape(a, &waves, &npad, &per_cu);
    if (waves == 4) {
        ctx->last_dense_kernel = "dense_mfma_big_w4_irls";
        return big_launch_t<4, 256, 2, 2, true>(ctx, g, grid);
    }
    ctx->last_dense_kernel = "dense_mfma_big_irls";
    return big_launch_t<8, 1024, 2, 2, true>(ctx, g, grid);
}

int dense_big_launch(gpc_ctx* ctx, const DenseArgs& a_in, int grid)
{
    DenseArgs a = a_in;
    double* v_star = a.v_star;
    const int ntw_ = (a.n_max + MF_TS - 1) / MF_TS;
    double* ws_alpha = nullptr;
    if (v_star) {
        // the variance kernel forms the mean from the same K* tiles: the fit predicts nothing, and leaves alpha behind
        ws_alpha = static_cast<double*>(ctx->ws) + big_slot_doubles(ntw_) * (size_t)a.P;
        a.v_star = nullptr;
        a.m = 0;
        if (!a.alpha_out) a.alpha_out = ws_alpha;
    }
    BigParams g;
    g.a = a;
    g.export_factor = v_star ? 1 : 0;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.ws = static_cast<double*>(ctx->ws);
    g.ntw = (a.n_max + MF_TS - 1) / MF_TS;
    g.slot = big_slot_doubles(g.ntw);
    int waves, npad, per_cu;
    big_shape(a, &waves, &npad, &per_cu);
    g.stamps = nullptr;
    g.irls_model = 0; g.irls_max_iter = 0; g.irls_tol = 0.0; g.irls_f_init = 0.0; g.irls_iters = nullptr; g.irls_fhat = nullptr;
    struct StampDump {
        gpc_ctx* ctx; unsigned long long* d; int P, waves;
        ~StampDump()
        {
            if (!d) return;
            unsigned long long h[BG_NPH * 8];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            static const char* names[BG_NPH] = {"load", "factorisation", "z gather", "backward", "predict", ""};
            fprintf(stderr, "[GPC_BIG_STAMPS] mean s_memtime ticks per patch, by wave\n");
            for (int q = 0; q < 5; ++q) {
                fprintf(stderr, "%-14s", names[q]);
                for (int w = 0; w < waves; ++w) fprintf(stderr, " %8.0f", (double)h[q * 8 + w] / P);
                fprintf(stderr, "\n");
            }
        }
    } dump{ctx, nullptr, a.P, waves};
    if (getenv("GPC_BIG_STAMPS")) {
        GPC_HIP(ctx, hipMalloc(&g.stamps, sizeof(unsigned long long) * BG_NPH * 8));
        GPC_HIP(ctx, hipMemsetAsync(g.stamps, 0, sizeof(unsigned long long) * BG_NPH * 8, ctx->stream));
        dump.d = g.stamps;
    }
    if (waves == 4) {
        ctx->last_dense_kernel = "dense_mfma_big_w4";
        return big_launch_t<4, 256, 2, 2>(ctx, g, grid);
    }
    ctx->last_dense_kernel = v_star ? "dense_mfma_big + dense_variance_big" : "dense_mfma_big";
    int rc = big_launch_t<8, 1024, 2, 2>(ctx, g, grid);
    if (rc != GPC_OK || !v_star) return rc;
    DenseArgs av = a;
    av.m = a_in.m;
    return dense_variance_big_launch(ctx, av, g.ntw, g.ws, g.slot, a.alpha_out,
                                     ws_alpha + (size_t)a.n_total * a.ny, v_star);
}
