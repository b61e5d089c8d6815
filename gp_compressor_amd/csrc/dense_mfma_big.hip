// dense_mfma_big.hip -- batched dense GP for 256 < n <= 1024 (BASELINE config 3: n = 512, config 5: n = 1024) on gfx950.
//
// Same computation as dense_mfma.hip / dense_generic.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45).  At these sizes the lower triangle of K (1 MB at n = 512, 4 MB at
// n = 1024) exceeds the register file of a CU, so the factor lives in a per-workgroup slot of a global workspace as
// 16 x 16 tiles in MFMA *operand image* layout (mfma_tile.h; 2 KB each, read back with two coalesced 16-byte loads per
// lane) and the factorisation is a tiled LEFT-looking Cholesky:
//
//   columns k .. k+3:  T_rc = A_rc - sum_{j<k} L_rj L_cj^T   (4 v_mfma_f64_16x16x4_f64 per (r, c, j)),   r = k .. nt-1
//              L_kk^-1 by the in-place Gauss-Jordan on the MFMA pipe (mf_diag_factor),   L_rk = T_rk L_kk^-T (4 MFMAs)
//
// * K itself is never stored: A_rk is evaluated (RBF + noise diagonal) straight into the accumulator registers when
//   column k starts.  HBM/L2 see the factor only: one write per tile, nt/3 reads on average.
// * FOUR tile columns per step (BG_C): every fetched tile L_rj feeds the accumulators of all four columns (the kernel streams
//   the factor from HBM / Infinity Cache: with one column per step it was plainly bound by that stream).  Tile rows are dealt
//   round-robin to the workers; a wave keeps up to BG_RMAX rows (x 4 columns) of accumulators, loads the shared column
//   operands once per j for all of them, and keeps the loads of the next j in flight.
// * The right-hand sides are a vector object: wave 0 solves z for the step's columns behind its chain (16 x 16 mat-vecs on the
//   VALU from the operand images, z in LDS).  (They used to ride along as one more tile row of MFMAs.)
// * Wave 0 owns the 4 x 4 diagonal block of the step -- its ten tiles are computed first (by all waves) and handed over; then the
//   chain: four diagonal factors, six TRSM / update pairs -- while the other waves stream their row updates; they poll an LDS
//   word for each L^-1 (bounded asm poll, mfma_tile.h).  One workgroup barrier per step.
// * The kernel is a template over waves per workgroup and padded size: <8, 1024> is the shape for 256 < n <= 1024 (one workgroup
//   per CU); <2, 256> -- chain wave + ONE worker, FOUR workgroups = four patches per CU, row passes taken from an LDS counter by
//   both waves -- is the C2 headline kernel (depth plane, 193 .. 256 points); <4, 256> (two workgroups per
//   CU) serves the colour planes under GPC_FORCE_BIG and is a cross-check.
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
    int32_t* ticket;              // patches beyond a workgroup's first one are taken from this counter (nullptr: blockIdx.x + k gridDim.x).  The
                                  // Newton loop takes 5 .. 11 solves per patch: with 16 patches per workgroup dealt in advance the slowest
                                  // workgroup of a C5 launch carried 11 % more solves than the average
    int export_factor;            // predictive variance (dense_variance.hip): the factor of EVERY patch stays in the workspace (slot =
                                  // patch, not workgroup) together with the L_kk^-1 images
    // IRLS instantiation only (BASELINE config 5, gpc_dense_irls_fit_predict): the Newton loop around the factorisation
    int irls_model;               // GPC_NOISE_PROBIT_REF / GPC_NOISE_PROBIT_STD
    int irls_max_iter;
    double irls_tol, irls_f_init;
    int32_t* irls_iters;          // [P] solves performed, or nullptr
    double* irls_fhat;            // [n_total] latent mode at the training points, or nullptr
};
// BG_HINT bit 1: the backward solve reads the factor once -- its loads carry the non-temporal hint, so that they do not push the block rows
// a step reads several times out of L2 (the one-wave kernel: 1.738 -> 1.680 ms with the same hint, dense_mfma_w1.hip)
// (C3 11.91 -> 11.78 ms, C5 299 -> 298 ms same box)
#ifndef BG_HINT
#define BG_HINT 1
#endif
#define BG_LOAD_BACK(p, l) ((BG_HINT & 1) ? mf_img_load_nt(p, l) : mf_img_load(p, l))
#define BG_NPH 12
// Diagnostic builds only (tools/r3_exp.sh; results are wrong by construction, timings tell what a phase costs under real overlap):
//   -DBG_EXP_HOT     every j-indexed operand load reads tile column 0 (L1 hits): what the factor stream costs
//   -DBG_EXP_NOGRAM  the Gram tiles are a constant diagonally dominant matrix (no exponentials)
//   -DBG_EXP_NOBACK / -DBG_EXP_NOPRED  skip the backward solve / the predictive mean
#ifdef BG_EXP_HOT
__device__ static __forceinline__ int bg_exp_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }   // (not hoistable)
#define BG_JX(j) bg_exp_zero()
#else
#define BG_JX(j) (j)
#endif
// -DBG_SUBSTAMPS (diagnostic build, tools/stamp_big.py): the factorisation step split into 5 diagonal-block tiles, 6 update loops,
// 7 waiting for an L^-1 (workers) or a hand-over (wave 0), 8 TRSMs (workers) or the chain (wave 0), 9 end-of-step barrier
#ifdef BG_SUBSTAMPS
#define BG_SUB(ph)                                                                                                   \
    do {                                                                                                             \
        if (g.stamps) {                                                                                              \
            const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                                          \
            sub_acc_[(ph) - 5] += t_now_ - t_sub_;                                                                   \
            t_sub_ = t_now_;                                                                                         \
        }                                                                                                            \
    } while (0)
#else
#define BG_SUB(ph) do { } while (0)
#endif
#define BG_STAMP(ph)                                                                                                 \
    do {                                                                                                             \
        if (g.stamps) {                                                                                              \
            const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                                          \
            if (lane == 0) atomicAdd(g.stamps + (ph) * 8 + wave, t_now_ - t_prev_);                                  \
            t_prev_ = t_now_;                                                                                        \
        }                                                                                                            \
    } while (0)

// LDS carve (doubles), for NPAD padded points and WAVES waves per workgroup:
//   exp table 64 | x0, x1 2 NPAD | z, w, alpha 3 NYP NPAD (NYP = 1: depth plane only, 3: depth + colour, or the IRLS vectors) | rsqrt row 32 | flags 16 | C L^-1 images | C (C-1)/2 images of the strictly
//   lower tiles of the step's diagonal block | hand-over of the C (C+1)/2 diagonal-block tiles (after the factorisation: the
//   predict accumulation buffer 4 x 256, ds_add_f64 targets)
__host__ __device__ constexpr int bg_hand_doubles(int c) { return c * (c + 1) / 2 * 256 > 1024 ? c * (c + 1) / 2 * 256 : 1024; }
// lalias (the two-wave shape): the images of the block's strictly lower tiles L_(k+i)(k+c) take the place of the hand-over tiles
// they were computed from (wave 0 reads T_ic, solves, stores L_ic over it) -- 12 KB less, which is what lets FOUR workgroups share a CU
__host__ __device__ constexpr int bg_lds_doubles(int npad, int c, int nyp, bool lalias = false)
{
    return 64 + (2 + 3 * nyp) * npad + 32 + 16 + c * 256 + (lalias ? 0 : (c * (c - 1) / 2) * 256) + bg_hand_doubles(c);
}

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
// The link functor of the IRLS instances, called OUT OF LINE: inlined, its ~60 polynomial constants (erfc, log) are materialised ahead of
// the Newton loop and then live -- as spills -- across the whole factorisation (476 B/lane of scratch in the 1024 instance; 144 with the
// call, C5 +3 %, round 4).  It runs n times per Newton iteration next to n^3/3 of factorisation work: the call costs nothing measurable.
struct bg_qr { double q, r; };
__device__ __attribute__((noinline)) static bg_qr bg_probit_call(int model, double noise, double y, double f)
{
    bg_qr o;
    gpc_probit_q_r(model, noise, y, f, 0.0, &o.q, &o.r);
    return o;
}
template <int BG_WAVES, int BG_NPAD, int BG_RMAX, int BG_OCC, bool BG_IRLS = false, int BG_C = 4, int BG_NYP = 3>
__global__ __launch_bounds__(BG_WAVES * 64, BG_OCC) void dense_big_kernel(BigParams g)
{
    static_assert(BG_RMAX == 1 || BG_RMAX == 2, "one copy of the update loop per possible row count");
    static_assert(BG_C == 4, "tile columns per step (the index maps of the diagonal block assume 4)");
    constexpr int BG_THREADS = BG_WAVES * 64;
    constexpr int NDT = BG_C * (BG_C + 1) / 2;           // tiles of the step's diagonal block
    constexpr bool LALIAS = BG_WAVES == 2;               // see bg_lds_doubles
    // Two-wave shape: the row passes of a step are not dealt but TAKEN -- both waves draw pass numbers from an LDS counter, the
    // worker from the start, wave 0 once its chain and the forward solve are done (with one worker the rows are the long pole of
    // a step and the chain wave would idle half of it at the barrier)
    constexpr bool STEAL = BG_WAVES == 2;
    static_assert(!STEAL || (BG_NPAD <= 512 && !BG_IRLS), "one pass counter per step in flag words 8 .. 15 (the IRLS loop keeps its step size there)");
    static_assert(BG_NYP == 3 || (BG_NYP == 1 && !BG_IRLS), "planes of the solve vectors (the IRLS loop keeps its vectors in planes 1, 2)");
    constexpr int B_PX0 = 64, B_PX1 = B_PX0 + BG_NPAD, B_ZV = B_PX1 + BG_NPAD, B_WV = B_ZV + BG_NYP * BG_NPAD, B_AV = B_WV + BG_NYP * BG_NPAD,
                  B_RS = B_AV + BG_NYP * BG_NPAD, B_FLAG = B_RS + 32, B_LINV = B_FLAG + 16, B_LBLK = B_LINV + BG_C * 256,
                  B_RED = B_LBLK + (LALIAS ? 0 : (BG_C * (BG_C - 1) / 2) * 256);
    static_assert(B_RED + bg_hand_doubles(BG_C) == bg_lds_doubles(BG_NPAD, BG_C, BG_NYP, LALIAS), "LDS carve");
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
    double* LinvC = lds + B_LINV;      // [BG_C][256]
    double* red = lds + B_RED;
    double* Hand = red;                // NDT x 256: hand-over of the diagonal-block tiles to wave 0 (the reduction buffer is idle then)
    // image of L_(k+i)(k+c), c < i: its own array at (i (i-1)/2 + c) * 256, or (LALIAS) in the slot of the hand-over tile (i, c)
    auto Lblk_at = [&](int i, int c) __attribute__((always_inline)) {
        return LALIAS ? Hand + (i * (i + 1) / 2 + c) * 256 : lds + B_LBLK + (i * (i - 1) / 2 + c) * 256;
    };
    // IRLS (ny == 1): planes 1, 2 of the solve vectors are free
    [[maybe_unused]] double* fv = zv + BG_NPAD;         // latent f
    [[maybe_unused]] double* dv = zv + 2 * BG_NPAD;     // W^-1/2
    [[maybe_unused]] double* sv = av + BG_NPAD;         // W^1/2
    [[maybe_unused]] double* tv = wv + BG_NPAD;         // working targets
    [[maybe_unused]] double* yl = wv + 2 * BG_NPAD;     // labels
    [[maybe_unused]] unsigned long long* delta_bits = reinterpret_cast<unsigned long long*>(lds + B_FLAG + 4);   // max |f_new - f| of the step, as bits
    int* hflag = flag + 16;            // [NDT]: step + 1 once diagonal-block tile d of the step is handed over
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
    const unsigned ready_addr = lds0 + (unsigned)(B_FLAG * 8 + 4);     // highest tile column whose L_kk^-1 is published
    const unsigned hf_addr = lds0 + (unsigned)(B_FLAG * 8 + 64);

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (BG_WAVES == 4) {
        // (four-wave shape, two workgroups per CU: both chain waves would sit on SIMD 0 -- the second workgroup's chain goes to SIMD 2)
        if (tid == 0) flag[27] = (int)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u);
        __syncthreads();
        wave ^= 2 * __builtin_amdgcn_readfirstlane(flag[27]);
    }
    if constexpr (BG_WAVES == 2) {
        // Two-wave shape, four workgroups per CU: the dispatcher puts the first waves of two workgroups on one SIMD and their second
        // waves on the next, so with fixed roles a SIMD would host two chain waves (its MFMA pipe mostly idle) and its neighbour two
        // workers (sharing one pipe).  The wave slot a workgroup's first wave got on its SIMD (HW_ID.WAVE_ID: 0 for the first
        // resident, 1 for the second) decides which of the two waves runs the chain: every SIMD then hosts one of each.  A
        // placement heuristic only -- any outcome is correct.
        if (tid == 0) flag[27] = (int)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u);     // hwreg(HW_REG_HW_ID, 0, 4)
        __syncthreads();
        wave ^= __builtin_amdgcn_readfirstlane(flag[27]);
    }
    const int lr = lane & 15, lg = lane >> 4;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny), m = A.m, ntw = g.ntw;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    double* Lt = g.ws + (size_t)blockIdx.x * g.slot;              // tiles (i, j): Lt + (i * ntw + j) * 256, i = 0 .. ntw
    double* LinvTg = Lt + (size_t)(ntw + 1) * ntw * MF_IMG;       // L_kk^-T images, k = 0 .. ntw-1
    double* LinvG = LinvTg + (size_t)ntw * MF_IMG;                // L_kk^-1 images (written when the factor is exported)

    gpc_exp_table_init(T);

    const int n_patches = A.sel ? __builtin_amdgcn_readfirstlane(A.sel_count[0]) : A.P;   // size-class dispatch: sel[0 .. count)
    int* const s_next_patch = flag + 30;      // (a free word of the flag block: static LDS on top of the 160 KB dynamic carve does not launch)
    auto next_patch = [&](int pk) {
        // (the Newton-loop instantiations only: in the others every patch of a class costs the same within a factor the static deal
        // averages out, and the two live values of the dynamic form spill in the 512-point shape)
        if (!BG_IRLS || !g.ticket) return pk + (int)gridDim.x;
        __syncthreads();
        if (tid == 0) *s_next_patch = (int)gridDim.x + atomicAdd(g.ticket, 1);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(*s_next_patch);
    };
    for (int pk = blockIdx.x; pk < n_patches; pk = next_patch(pk)) {
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
            if constexpr (BG_IRLS) {
                if (tid == 0 && g.irls_iters) g.irls_iters[patch] = 0;     // no Newton step was taken
            }
            continue;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);
        unsigned long long t_prev_ = g.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        [[maybe_unused]] unsigned long long t_sub_ = t_prev_;
        [[maybe_unused]] unsigned long long sub_acc_[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
        unsigned long long* ext_bits = reinterpret_cast<unsigned long long*>(lds + B_FLAG + 3);   // max-norm extent, as bits
        if (tid < NDT) hflag[tid] = 0;
        if (STEAL && tid < 8) flag[8 + tid] = 0;               // pass counters of the (up to eight) steps
        if (tid == 0) {
            flag[0] = 0;
            flag[1] = -1;
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
            for (int c = 0; c < BG_NYP; ++c) { wv[c * BG_NPAD + i] = 0.0; av[c * BG_NPAD + i] = 0.0; }   // av: no stale LDS beyond the solved rows
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

        // ---- tiled left-looking Cholesky, BG_C = 4 tile columns (k .. k+3) per step ----
        // Four columns per step quarter the dominant stream: every tile L_rj fetched for the update of row r feeds the accumulators
        // of all four columns (16 MFMAs per 2 KB).  Wave 0 owns the 4 x 4 diagonal block: factor (k,k), then row by row TRSM /
        // update / factor, publishing each L^-1 as it appears.  The rows r >= k+4 are dealt to the workers (or taken from a counter,
        // two-wave shape): update the four accumulators, then column by column  acc_c -= sum_{c2<c} L_r(k+c2) L_(k+c)(k+c2)^T,
        // L_r(k+c) = TRSM(acc_c).
#ifdef BG_EXP_NOGRAM
#define BG_GRAM_VALUE(xi0_, xi1_, pj_) ((pi_ == (pj_)) ? sf : 0.001 * sf)
#else
#define BG_GRAM_VALUE(xi0_, xi1_, pj_) (small_gram ? gpc_rbf_small(sf, cexp, xi0_, xi1_, px0[pj_], px1[pj_]) : gpc_rbf_neg(sf, cexp, xi0_, xi1_, px0[pj_], px1[pj_], T))
#endif
#define BG_INIT_TILE(dst, r_, kc_)                                                                                   \
    do {                                                                                                             \
        if ((r_) < nt) {                                                                                             \
            const int pi_ = MF_TS * (r_) + lr;                                                                       \
            const double xi0_ = px0[pi_], xi1_ = px1[pi_];                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
                const int pj_ = MF_TS * (kc_) + lg + 4 * q_;                                                         \
                double v_ = BG_GRAM_VALUE(xi0_, xi1_, pj_);                                                          \
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
        } else {   /* beyond the last tile row: nothing (the right-hand sides are a vector object, see the forward solve) */ \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) (dst)[q_] = 0.0;                                        \
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
                    const bg_qr qr_ = bg_probit_call(g.irls_model, noise, yl[i], fv[i]);
                    const double q_ = qr_.q, r_ = qr_.r;
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
            if (tid < NDT) hflag[tid] = 0;
            if (tid == 0) {
                flag[0] = 0;
                flag[1] = -1;
                *delta_bits = 0ull;
            }
            __syncthreads();
            nan_w = flag[4] != 0;
            if (nan_w) break;
        }
        int step = 0;
        for (int k = 0; k < nt; k += BG_C, ++step) {
            const int nc = min(BG_C, nt - k);                              // tile columns of this step
            constexpr int NWK = BG_WAVES - 1;
            // ---- the diagonal block first: its nc (nc + 1) / 2 tiles, one per wave (tile 0 = (k, k) on wave 0 itself) ----
            // Tile d = i (i + 1) / 2 + c is T_(k+i)(k+c) = A - sum_{j<k} L_(k+i)j L_(k+c)j^T; wave 0 needs them in the order of d.
            // Nothing else is live here, so eight (j) operand pairs are in flight per wave.
            d4 Dmine = d4{0.0, 0.0, 0.0, 0.0};
            BG_SUB(9);
            {
                constexpr int TPW_ALL = (NDT - 1 + NWK - 1) / NWK;         // tiles per worker, at most (wave 0: tile 0 only)
                constexpr int DPW = TPW_ALL < 3 ? TPW_ALL : 3;             // ... side by side; the rest in further batches (one worker: 3 x 3)
                static_assert(DPW >= 1, "diagonal-block tiles per worker");
#define BG_DG_LOAD(st, j0, NPC)                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int jq_ = 0; jq_ < 4 / NPC; ++jq_) {                                                  \
            const int jj_ = min((j0) + jq_, kl);                                                                     \
            _Pragma("unroll") for (int t = 0; t < NPC; ++t) {                                                        \
                ga[st][jq_ * NPC + t] = mf_img_load(ra[t] + (size_t)BG_JX(jj_) * MF_IMG, lane);                             \
                gb[st][jq_ * NPC + t] = mf_img_load(rb[t] + (size_t)BG_JX(jj_) * MF_IMG, lane);                             \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define BG_DG_USE(st, NPC)                                                                                           \
    do {                                                                                                             \
        _Pragma("unroll") for (int jq_ = 0; jq_ < 4 / NPC; ++jq_)                                                    \
            _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                          \
                dacc[t] = bg_mfma4_neg(ga[st][jq_ * NPC + t], gb[st][jq_ * NPC + t], dacc[t]);                       \
    } while (0)
                // NPC tiles side by side, 4 / NPC values of j per operand stage, two stages; k is a multiple of BG_C = 4.  The
                // first loads are issued before the Gram tiles are evaluated.
#define BG_DG_TILES(NPC)                                                                                             \
    do {                                                                                                             \
        constexpr int JC_ = 4 / NPC;                                                                                 \
        if (k > 0) BG_DG_LOAD(0, 0, NPC);                                                                            \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t) {                                                            \
            const int bi_ = dd[t] >= 6 ? 3 : dd[t] >= 3 ? 2 : dd[t] >= 1 ? 1 : 0;                                    \
            BG_INIT_TILE(dacc[t], k + bi_, k + dd[t] - bi_ * (bi_ + 1) / 2);                                         \
        }                                                                                                            \
        for (int j = 0; j < k; j += 2 * JC_) {                                                                       \
            BG_DG_LOAD(1, j + JC_, NPC);                                                                             \
            BG_DG_USE(0, NPC);                                                                                       \
            BG_DG_LOAD(0, j + 2 * JC_, NPC);                                                                         \
            if (j + JC_ < k) BG_DG_USE(1, NPC);                                                                      \
        }                                                                                                            \
    } while (0)
                if (NWK == 1) {
                    // Two-wave shape: ONE sweep over j per wave, the row operands L_(k+i)j fetched once per j for every product they
                    // feed (tile by tile it was two loads per tile and j), two stages.  Wave 0 takes the three tiles its chain starts
                    // with -- (0,0), (1,0), (1,1): rows k, k+1 -- so that the first two diagonal factors do not wait for the worker;
                    // the worker takes the seven tiles of the block rows 2 and 3.
                    const int d_lo = (wave == 0) ? 0 : 3, d_hi = (wave == 0) ? 2 : NDT - 1;
                    const int i_hi = (wave == 0) ? 1 : BG_C - 1;
                    const int kl = k - 1;
                    const double* rrow[BG_C];
#pragma unroll
                    for (int i = 0; i < BG_C; ++i) rrow[i] = Lt + ((size_t)(k + min(i, nc - 1)) * ntw) * MF_IMG;
                    d4 op[2][BG_C], tacc[NDT];
#pragma unroll
                    for (int i = 0; i < BG_C; ++i) op[0][i] = op[1][i] = d4{0.0, 0.0, 0.0, 0.0};
                    if (k > 0) {
#pragma unroll
                        for (int i = 0; i < BG_C; ++i)
                            if (i <= i_hi) op[0][i] = mf_img_load(rrow[i], lane);
                    }
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const int bi = d >= 6 ? 3 : d >= 3 ? 2 : d >= 1 ? 1 : 0, bc = d - bi * (bi + 1) / 2;
                        tacc[d] = d4{0.0, 0.0, 0.0, 0.0};
                        if (d >= d_lo && d <= d_hi && bi < nc) BG_INIT_TILE(tacc[d], k + bi, k + bc);
                    }
                    for (int j = 0; j < k; j += 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int jn = min(j + h + 1, kl);
#pragma unroll
                            for (int i = 0; i < BG_C; ++i)
                                if (i <= i_hi) op[h ^ 1][i] = mf_img_load(rrow[i] + (size_t)BG_JX(jn) * MF_IMG, lane);
#pragma unroll
                            for (int d = 0; d < NDT; ++d) {
                                const int bi = d >= 6 ? 3 : d >= 3 ? 2 : d >= 1 ? 1 : 0, bc = d - bi * (bi + 1) / 2;
                                if (d >= d_lo && d <= d_hi && bi < nc) tacc[d] = bg_mfma4_neg(op[h][bc], op[h][bi], tacc[d]);
                            }
                        }
                    }
                    Dmine = tacc[0];                               // (wave 0; the worker's copy is zero and unused)
#pragma unroll
                    for (int d = 1; d < NDT; ++d) {
                        const int bi = d >= 6 ? 3 : d >= 3 ? 2 : 1;
                        if (d >= d_lo && d <= d_hi && bi < nc) {
                            *reinterpret_cast<d4*>(Hand + d * 256 + mf_opaque(lane) * 4) = tacc[d];
                            mf_publish(hflag + d, step + 1);
                        }
                    }
                } else
                for (int tb = 0; tb < TPW_ALL; tb += DPW) {
                int dd[DPW], nd = 0;
                const double *ra[DPW], *rb[DPW];
                d4 dacc[DPW];
#pragma unroll
                for (int t = 0; t < DPW; ++t) {
                    const int d = (wave == 0) ? (tb + t == 0 ? 0 : NDT) : wave + NWK * (tb + t);
                    const int bi = d >= 6 ? 3 : d >= 3 ? 2 : d >= 1 ? 1 : 0;
                    const bool on = d < NDT && bi < nc;                    // (monotone in t: the valid tiles come first)
                    const int dv_ = on ? d : 0, bi_ = on ? bi : 0, bc_ = dv_ - bi_ * (bi_ + 1) / 2;
                    dd[t] = dv_;
                    nd += on ? 1 : 0;
                    ra[t] = Lt + ((size_t)(k + bc_) * ntw) * MF_IMG;
                    rb[t] = Lt + ((size_t)(k + bi_) * ntw) * MF_IMG;
                }
                if (nd == 0) break;
                const int kl = k - 1;
                d4 ga[2][4], gb[2][4];
#pragma unroll
                for (int t = 0; t < DPW; ++t) dacc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int q_ = 0; q_ < 4; ++q_) ga[st][q_] = gb[st][q_] = d4{0.0, 0.0, 0.0, 0.0};
                if (nd == 1) BG_DG_TILES(1);
                if constexpr (DPW >= 2) { if (nd == 2) BG_DG_TILES(2); }
                if constexpr (DPW >= 3) { if (nd == 3) BG_DG_TILES(3); }
                if (wave == 0) {
                    Dmine = dacc[0];
                } else {
#pragma unroll
                    for (int t = 0; t < DPW; ++t) {
                        if (t < nd) {
                            *reinterpret_cast<d4*>(Hand + dd[t] * 256 + lane * 4) = dacc[t];
                            mf_publish(hflag + dd[t], step + 1);
                        }
                    }
                }
                }
            }
            BG_SUB(5);
            if (wave == 0) {
                // the serial chain of the block: factor (k,k); then row by row  L_ic = (T_ic - sum_{c2<c} L_ic2 L_cc2^T) L_cc^-T,
                // T_ii -= sum_c L_ic L_ic^T, factor -- every L^-1 published as it appears, the block's tiles left in LDS for the workers
                bool ok = mf_diag_factor<true>(Dmine, rsbuf, LinvC, LinvTg + (size_t)k * MF_IMG, g.pivot_tol);
#ifdef BG_EXP_HOT
                ok = true;
#endif
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (g.export_factor) mf_img_store(LinvG + (size_t)k * MF_IMG, lane, mf_img_load(LinvC, lane));
                if (!ok && lane == 0) flag[0] = 1;
                if (ok) mf_publish(ready, k);
#pragma unroll
                for (int i = 1; i < BG_C; ++i) {
                    if (ok && i < nc) {
                        d4 Lrow[BG_C - 1];
#pragma unroll
                        for (int c = 0; c < i; ++c) {
                            // (LDS addresses from an opaque lane id: hoisted out of the step loop they end up in scratch)
                            const int ln = mf_opaque(lane);
                            BG_SUB(8);
                            timed_out |= !mf_wait_ge(hf_addr + 4u * (unsigned)(i * (i + 1) / 2 + c), step + 1);
                            BG_SUB(7);
                            d4 Tt = *reinterpret_cast<const d4*>(Hand + (i * (i + 1) / 2 + c) * 256 + ln * 4);
#pragma unroll
                            for (int c2 = 0; c2 < c; ++c2)
                                Tt = bg_mfma4_neg(mf_img_load(Lblk_at(c, c2), ln), Lrow[c2], Tt);
                            const d4 lvc = mf_img_load(LinvC + c * 256, ln);
                            Lrow[c] = BG_TRSM(lvc, Tt);                     // operand image of L_(k+i)(k+c)
                            mf_img_store(Lt + ((size_t)(k + i) * ntw + k + c) * MF_IMG, ln, Lrow[c]);
                            mf_img_store(Lblk_at(i, c), ln, Lrow[c]);
                        }
                        BG_SUB(8);
                        timed_out |= !mf_wait_ge(hf_addr + 4u * (unsigned)(i * (i + 1) / 2 + i), step + 1);
                        BG_SUB(7);
                        d4 Dii = *reinterpret_cast<const d4*>(Hand + (i * (i + 1) / 2 + i) * 256 + mf_opaque(lane) * 4);
#pragma unroll
                        for (int c = 0; c < i; ++c) Dii = bg_mfma4_neg(Lrow[c], Lrow[c], Dii);
                        ok = mf_diag_factor<true>(Dii, rsbuf, LinvC + i * 256, LinvTg + (size_t)(k + i) * MF_IMG, g.pivot_tol);
#ifdef BG_EXP_HOT
                        ok = true;
#endif
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                                if (g.export_factor) mf_img_store(LinvG + (size_t)(k + i) * MF_IMG, lane, mf_img_load(LinvC + i * 256, mf_opaque(lane)));
                        if (!ok && lane == 0) flag[0] = 1;
                        if (ok) mf_publish(ready, k + i);
                    }
                }
                mf_publish(ready, k + nc - 1);     // (also on failure: the workers leave their waits and see the flag)
                BG_SUB(8);
                // ---- forward solve of the step's columns: z_(k+c) = L_cc^-1 (y_(k+c) - sum_{j<k} L_(k+c)j z_j - sum_{c2<c} L_(k+c)(k+c2) z_(k+c2)) ----
                // A vector object: per tile one 16 x 16 mat-vec on the VALU (4 FMAs per lane and channel) against 16 MFMAs when the
                // right-hand sides were a tile row.  The operand image of M gives lane l the entries M[l & 15][(l >> 4) + 4 s]: the
                // lane's partial sum over its four columns, then the four lane groups of a row are added.  z lives in LDS (zv).
                if (ok) {
                    const int ln = mf_opaque(lane), row = ln & 15, grp = ln >> 4;
                    double part[BG_C][BG_NYP];
#pragma unroll
                    for (int c = 0; c < BG_C; ++c)
#pragma unroll
                        for (int ch = 0; ch < BG_NYP; ++ch) part[c][ch] = 0.0;
                    // the rows k .. k+nc-1 of the factor, columns j < k: in the workspace since the earlier steps (the workers stream the
                    // same tiles as their column operands right now: L1 / L2 hits); one j ahead in flight
                    const double* rc_[BG_C];
#pragma unroll
                    for (int c = 0; c < BG_C; ++c) rc_[c] = Lt + ((size_t)(k + min(c, nc - 1)) * ntw) * MF_IMG;
                    d4 fa[2][BG_C];
#pragma unroll
                    for (int c = 0; c < BG_C; ++c) fa[0][c] = fa[1][c] = d4{0.0, 0.0, 0.0, 0.0};
                    if (k > 0) {
#pragma unroll
                        for (int c = 0; c < BG_C; ++c) fa[0][c] = mf_img_load(rc_[c], ln);
                    }
                    for (int j = 0; j < k; j += 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int jn = min(j + h + 1, k - 1);
#pragma unroll
                            for (int c = 0; c < BG_C; ++c) fa[h ^ 1][c] = mf_img_load(rc_[c] + (size_t)BG_JX(jn) * MF_IMG, ln);
                            if (j + h < k) {
#pragma unroll
                                for (int ch = 0; ch < BG_NYP; ++ch) {
                                    if (ch < ny) {
                                        const double* zq = zv + ch * BG_NPAD + MF_TS * (j + h) + grp;
                                        const double z0 = zq[0], z1 = zq[4], z2 = zq[8], z3 = zq[12];
#pragma unroll
                                        for (int c = 0; c < BG_C; ++c)
                                            part[c][ch] += (fa[h][c][0] * z0 + fa[h][c][1] * z1) + (fa[h][c][2] * z2 + fa[h][c][3] * z3);
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < BG_C; ++c) {
                        if (c < nc) {
                            const int pj = MF_TS * (k + c) + row;
#pragma unroll
                            for (int ch = 0; ch < BG_NYP; ++ch) {
                                if (ch < ny) {
                                    // within the block: the tiles of the step's diagonal block are in LDS (Lblk), as are the z of its earlier columns
                                    double pp = part[c][ch];
#pragma unroll
                                    for (int c2 = 0; c2 < c; ++c2) {
                                        const d4 lb = mf_img_load(Lblk_at(c, c2), ln);
                                        const double* zq = zv + ch * BG_NPAD + MF_TS * (k + c2) + grp;
                                        pp += (lb[0] * zq[0] + lb[1] * zq[4]) + (lb[2] * zq[8] + lb[3] * zq[12]);
                                    }
                                    pp += __shfl_xor(pp, 16, 64);
                                    pp += __shfl_xor(pp, 32, 64);
                                    double yv_;
                                    if constexpr (BG_IRLS) yv_ = (pj < n) ? sv[pj] * tv[pj] : 0.0;
                                    else yv_ = (pj < n) ? A.y[(size_t)ch * A.n_total + o + pj] : 0.0;
                                    if (grp == 0) zv[ch * BG_NPAD + pj] = yv_ - pp;          // t_c, where z_c goes next
                                    __builtin_amdgcn_wave_barrier();
                                    const d4 lv = mf_img_load(LinvC + c * 256, ln);
                                    const double* tq = zv + ch * BG_NPAD + MF_TS * (k + c) + grp;
                                    double zz = (lv[0] * tq[0] + lv[1] * tq[4]) + (lv[2] * tq[8] + lv[3] * tq[12]);
                                    zz += __shfl_xor(zz, 16, 64);
                                    zz += __shfl_xor(zz, 32, 64);
                                    __builtin_amdgcn_wave_barrier();
                                    if (grp == 0) zv[ch * BG_NPAD + pj] = zz;
                                    __builtin_amdgcn_wave_barrier();
                                }
                            }
                        }
                    }
                }
                BG_SUB(8);
            }
            if (STEAL || wave != 0) {
                // Rows k + nc .. nt - 1 are dealt round-robin to the workers, the wave that shares its SIMD with wave 0 first, the
                // ones with two diagonal-block tiles last.  (The right-hand sides used to ride along as one more tile row -- a full
                // row of MFMAs for 1..3 live channels of 16, and with 4 m + 1 rows per step the busiest SIMD always carried m + 1
                // where m + 1/4 was its share; they are a vector object now, solved by wave 0 behind its chain: C3 13.1 -> 12.7 ms.
                // Dealing SIMD by SIMD instead -- wave 4, alone with the chain wave on its SIMD, taking a whole SIMD's share in more
                // passes -- was measured and is slower, 13.3 ms: one wave does not keep its pipe as busy as two that interleave.)
                const int q0 = (NWK == 7) ? ((wave == 4) ? 0 : (wave >= 5) ? 8 - wave : 7 - wave) : NWK - wave;
                const int rows_tot = nt - (k + nc);
                const int rows_w = (rows_tot - 1 - q0 >= 0) ? (rows_tot - 1 - q0) / NWK + 1 : 0;
                bool stop = false;
                for (int p0 = 0; !stop; p0 += BG_RMAX) {
                    int np, first_row, row_stride;
                    if constexpr (STEAL) {
                        int pnum = 0;
                        if (lane == 0) pnum = __hip_atomic_fetch_add(flag + 8 + step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        pnum = __builtin_amdgcn_readfirstlane(pnum);
                        first_row = pnum * BG_RMAX;
                        if (first_row >= rows_tot) break;
                        np = min(BG_RMAX, rows_tot - first_row);
                        row_stride = 1;
                    } else {
                        if (p0 >= rows_w) break;
                        np = min(BG_RMAX, rows_w - p0);
                        first_row = q0 + NWK * p0;
                        row_stride = NWK;
                    }
                    d4 acc[BG_C][BG_RMAX];
                    int rr[BG_RMAX];
                    // two operand stages: while one feeds the 4 BG_C NPC MFMAs of a j, the loads of the next j are in flight.
                    // The loads are UNCONDITIONAL (column index clamped into the block, j clamped to k-1: redundant re-reads):
                    // under runtime conditions hipcc cannot count the outstanding loads and falls back to s_waitcnt vmcnt(0)
                    // in every iteration, i.e. no prefetch at all.  k is even: no tail.  The first stage is requested before the
                    // Gram tiles are evaluated (the factor comes from HBM / Infinity Cache, thousands of cycles away).
                    const double* rc_[BG_C];
#pragma unroll
                    for (int c = 0; c < BG_C; ++c) rc_[c] = Lt + ((size_t)(k + min(c, nc - 1)) * ntw) * MF_IMG;
                    const double* rw_[BG_RMAX];
#pragma unroll
                    for (int t = 0; t < BG_RMAX; ++t) {
                        rr[t] = k + nc + first_row + row_stride * min(t, np - 1);   // (t >= np: a copy of the last row, never stored)
                        rw_[t] = Lt + ((size_t)rr[t] * ntw) * MF_IMG;
                    }
                    d4 sa[2][BG_C], sb[2][BG_RMAX];
                    const int kl = k - 1;
#define BG_LOAD_STAGE(st, jj, NPC)                                                                                   \
    do {                                                                                                             \
        _Pragma("unroll") for (int c = 0; c < BG_C; ++c) sa[st][c] = mf_img_load(rc_[c] + (size_t)BG_JX(jj) * MF_IMG, lane); \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t) sb[st][t] = mf_img_load(rw_[t] + (size_t)BG_JX(jj) * MF_IMG, lane);  \
    } while (0)
#define BG_USE_STAGE(st, NPC)                                                                                        \
    do {                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                              \
            _Pragma("unroll") for (int c = 0; c < BG_C; ++c) acc[c][t] = bg_mfma4_neg(sa[st][c], sb[st][t], acc[c][t]); \
    } while (0)
#define BG_UPDATE_LOOP(NPC)                                                                                          \
    do {                                                                                                             \
        if (k > 0) BG_LOAD_STAGE(0, 0, NPC);                                                                         \
        _Pragma("unroll") for (int t = 0; t < NPC; ++t)                                                              \
            _Pragma("unroll") for (int c = 0; c < BG_C; ++c)                                                         \
                if (c < nc) BG_INIT_TILE(acc[c][t], rr[t], k + c);                                                   \
        for (int j = 0; j < k; j += 2) {                                                                             \
            BG_LOAD_STAGE(1, j + 1, NPC);                                                                            \
            BG_USE_STAGE(0, NPC);                                                                                    \
            BG_LOAD_STAGE(0, min(j + 2, kl), NPC);                                                                   \
            BG_USE_STAGE(1, NPC);                                                                                    \
        }                                                                                                            \
    } while (0)
#pragma unroll
                    for (int t = 0; t < BG_RMAX; ++t)
#pragma unroll
                        for (int c = 0; c < BG_C; ++c) acc[c][t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
#pragma unroll
                        for (int c = 0; c < BG_C; ++c) sa[st][c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int t = 0; t < BG_RMAX; ++t) sb[st][t] = d4{0.0, 0.0, 0.0, 0.0};
                    }
                    if constexpr (BG_RMAX == 2) {
                        if (np == 2) BG_UPDATE_LOOP(2);
                        else BG_UPDATE_LOOP(1);
                    } else {
                        BG_UPDATE_LOOP(1);
                    }
                    BG_SUB(6);
                    // column by column as the inverses appear:  T_r(k+c) -= sum_{c2<c} L_r(k+c2) L_(k+c)(k+c2)^T,  L_r(k+c) = T L_cc^-T
#pragma unroll
                    for (int c = 0; c < BG_C; ++c) {
                        if (c < nc && !stop) {
                            timed_out |= !mf_wait_ge(ready_addr, k + c);
                            BG_SUB(7);
                            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                stop = true;
                            } else {
                                const int ln = mf_opaque(lane);     // (see wave 0)
                                const d4 lv = mf_img_load(LinvC + c * 256, ln);
#pragma unroll
                                for (int t = 0; t < BG_RMAX; ++t) {
                                    if (t < np) {
#pragma unroll
                                        for (int c2 = 0; c2 < c; ++c2)
                                            acc[c][t] = bg_mfma4_neg(mf_img_load(Lblk_at(c, c2), ln), acc[c2][t], acc[c][t]);
                                        acc[c][t] = BG_TRSM(lv, acc[c][t]);
                                        mf_img_store(Lt + ((size_t)rr[t] * ntw + k + c) * MF_IMG, ln, acc[c][t]);
                                    }
                                }
                                BG_SUB(8);
                            }
                        }
                    }
                }
            }
            __syncthreads();   // the column block is in the workspace; the L^-1 images and the hand-over tiles may be overwritten
            bad = flag[0] != 0;
            if (bad) break;
        }
        if (bad) break;
        BG_STAMP(1);
#ifdef BG_SUBSTAMPS
        if (g.stamps && lane == 0)
            for (int q_ = 0; q_ < 5; ++q_) atomicAdd(g.stamps + (5 + q_) * 8 + wave, sub_acc_[q_]);
#endif
        __syncthreads();   // z (LDS, written by wave 0 step by step) is complete
        BG_STAMP(2);
#ifndef BG_EXP_NOBACK
        // ---- backward solve L^T alpha = z, tile columns from the last to the first ----
        // Column k needs the tiles (i, k), i = k+1+wave+W t, and L_kk^-T: they do not depend on alpha, so the loads of
        // column k-1 are issued before the products of column k (the factor sits in HBM / Infinity Cache, ~2k cycles away).
        {
            constexpr int BT = (BG_NPAD / MF_TS + BG_WAVES - 1) / BG_WAVES;      // tiles per wave and column, at most
            d4 cur[BT], nxt[BT], lt_cur = d4{0.0, 0.0, 0.0, 0.0}, lt_nxt = lt_cur;
#pragma unroll
            for (int t = 0; t < BT; ++t) cur[t] = nxt[t] = d4{0.0, 0.0, 0.0, 0.0};
            if (wave == 0) lt_cur = BG_LOAD_BACK(LinvTg + (size_t)(nt - 1) * MF_IMG, lane);
            for (int k = nt - 1; k >= 0; --k) {
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < BT; ++t) {
                        const int i = k + wave + BG_WAVES * t;               // rows of column k-1: i >= k
                        if (i < nt) nxt[t] = BG_LOAD_BACK(Lt + ((size_t)i * ntw + (k - 1)) * MF_IMG, lane);
                    }
                    if (wave == 0) lt_nxt = BG_LOAD_BACK(LinvTg + (size_t)(k - 1) * MF_IMG, lane);
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
#endif
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
#ifndef BG_EXP_NOPRED
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
#endif
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
static void big_shape(const DenseArgs& a, bool irls, int* waves, int* npad, int* per_cu)
{
    // depth plane, n <= 256: TWO waves per workgroup (the chain wave + one worker) and FOUR workgroups per CU (40 KB of LDS each)
    if (a.n_max <= 256 && a.ny == 1 && !irls && !a.v_star && !getenv("GPC_BIG_NO_W2")) { *waves = 2; *npad = 256; *per_cu = 4; }
    else if (a.n_max <= 256) { *waves = 4; *npad = 256; *per_cu = 2; }
    // (the two-wave shape at 512 points, three workgroups per CU at 50 KB of LDS: 16.0 ms on C3 against 12.2 -- six waves per CU, and
    // one worker cannot carry a step's 28 row passes)
    // depth plane only, up to 512 points: four waves, two patches per CU (62 KB of LDS each).  Measured on the producer's own batches
    // (273 .. 324 points): GP phase 3.29 against 3.42 ms.  At n = 512 (C3) the 8-wave shape used to win, 13.2 against 13.4 ms -- both
    // chain waves sat on SIMD 0, which then idled; with the second workgroup's chain on SIMD 2 (HW_ID wave slot, see the kernel) the
    // two-workgroup shape wins, 12.23 against 12.63 ms on the same box (round 3)
    else if (a.n_max <= (getenv("GPC_BIG_W4_384") ? 384 : 512) && a.ny == 1 && !irls && !getenv("GPC_BIG_NO_W4")) { *waves = 4; *npad = 512; *per_cu = 2; }
    else { *waves = 8; *npad = 1024; *per_cu = 1; }
}

size_t dense_big_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out)
{
    int waves, npad, per_cu;
    big_shape(a, false, &waves, &npad, &per_cu);
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

template <int W, int NP, int RM, int OC, bool IRLS = false, int NYP = 3>
static int big_launch_t(gpc_ctx* ctx, const BigParams& g, int grid)
{
    const size_t lds = sizeof(double) * (size_t)bg_lds_doubles(NP, 4, NYP, W == 2);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_big_kernel<W, NP, RM, OC, IRLS, 4, NYP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((dense_big_kernel<W, NP, RM, OC, IRLS, 4, NYP>), dim3(grid), dim3(W * 64), lds, ctx->stream, g);
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
    if (!ctx->tickets) GPC_HIP(ctx, hipMalloc(&ctx->tickets, 64 * sizeof(int32_t)));
    GPC_HIP(ctx, hipMemsetAsync(ctx->tickets, 0, sizeof(int32_t), ctx->stream));
    g.ticket = getenv("GPC_BIG_STATIC") ? nullptr : ctx->tickets;
    int waves, npad, per_cu;
    big_shape(a, true, &waves, &npad, &per_cu);
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
    big_shape(a, false, &waves, &npad, &per_cu);
    g.stamps = nullptr;
    g.irls_model = 0; g.irls_max_iter = 0; g.irls_tol = 0.0; g.irls_f_init = 0.0; g.irls_iters = nullptr; g.irls_fhat = nullptr;
    g.ticket = nullptr;
    struct StampDump {
        gpc_ctx* ctx; unsigned long long* d; int P, waves;
        ~StampDump()
        {
            if (!d) return;
            unsigned long long h[BG_NPH * 8];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            static const char* names[BG_NPH] = {"load", "factorisation", "z gather", "backward", "predict", "- diag tiles", "- update",
                                                "- wait", "- trsm/chain", "- barrier", "", ""};
            fprintf(stderr, "[GPC_BIG_STAMPS] mean s_memtime ticks per patch, by wave\n");
            for (int q = 0; q < 10; ++q) {
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
    if (waves == 2) {
        ctx->last_dense_kernel = "dense_mfma_big_w2";
        return big_launch_t<2, 256, 2, 2, false, 1>(ctx, g, grid);
    }
    if (waves == 4 && npad == 256) {
        ctx->last_dense_kernel = "dense_mfma_big_w4";
        return big_launch_t<4, 256, 2, 2>(ctx, g, grid);
    }
    int rc;
    if (waves == 4) {
        ctx->last_dense_kernel = v_star ? "dense_mfma_big + dense_variance_big" : "dense_mfma_big";      // (the shape is not part of the name)
        rc = big_launch_t<4, 512, 2, 2, false, 1>(ctx, g, grid);
    } else {
        ctx->last_dense_kernel = v_star ? "dense_mfma_big + dense_variance_big" : "dense_mfma_big";
        rc = big_launch_t<8, 1024, 2, 2>(ctx, g, grid);
    }
    if (rc != GPC_OK || !v_star) return rc;
    DenseArgs av = a;
    av.m = a_in.m;
    return dense_variance_big_launch(ctx, av, g.ntw, g.ws, g.slot, a.alpha_out,
                                     ws_alpha + (size_t)a.n_total * a.ny, v_star);
}
