// dense_mfma.hip -- register-resident batched dense GP for n <= 256 on gfx950 (the bench path, BASELINE config 2).
//
// Same computation as dense_generic.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45), restructured for CDNA4:
//
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per patch; the whole lower triangle of K -- 136 tiles of
//     16 x 16 doubles = 272 KB, more than the 160 KB of LDS -- lives in the VGPR file as
//     v_mfma_f64_16x16x4_f64 accumulators (17 tiles = 136 VGPRs per wave).  K never touches HBM or L2.
//   * right-looking tiled Cholesky: per tile column k the owner of the diagonal tile inverts its Cholesky factor
//     in registers ([A | I] -> [L^T | L^-1] with cross-lane v_readlane broadcasts), the panel TRSM is
//     L_ik^T = L_kk^-1 * A_ik^T as 4 MFMAs per tile, the trailing update A_ij -= L_ik L_jk^T is 4 MFMAs per tile
//     with both operands read from a 32 KB LDS panel in "operand layout" (lane l holds row l&15, k = (l>>4)+4s:
//     one conflict-free 32-byte read per operand).
//   * tiles are stored TRANSPOSED (T_ij = A_ij^T in the MFMA C/D layout), which makes the accumulator registers of
//     a panel tile directly the B operand of its TRSM and the TRSM result directly the operand-layout image of
//     L_ik: no cross-lane movement anywhere in the O(n^3) part.
//   * the diagonal factorisation of column k+1 is issued by its owner right after that tile's step-k update, so it
//     overlaps the other waves' MFMA work; the forward solve rides along with the TRSM, the backward solve walks
//     the register-resident factor, and the predictive mean on the sz x sz decompression grid is evaluated
//     separably (K* = Ex o Ey, /root/reference/src/gp_compressor.cpp:317-332) as 4 more MFMA tiles per wave.
//
// Lane maps (verified by tools/probe_mfma_f64.hip): A operand lane l = A[l&15][l>>4], B operand lane l =
// B[l>>4][l&15], C/D register r of lane l = D[(l>>4) + 4r][l&15].
#include <vector>

#include "gpc_device.h"
#include "gpc_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MF_THREADS 512
#define MF_WAVES 8
#define MF_TS 16
#define MF_NPAD 256

struct MfmaParams {
    DenseArgs a;
    double c_exp;
    double pivot_tol;               // a pivot <= pivot_tol is reported as GPC_STATUS_NOT_SPD
    unsigned long long* stamps;     // diagnostic build (-DMF_STAMPS) only: [block][wave][MF_NPH] cycle sums
};

// Diagnostic phase timing (cdna_hip_programming.md section 7, "In-kernel stamps"): compiled in only with -DMF_STAMPS
// (tools/stamp_mfma.py builds a separate library); the shipped kernel executes no stamp.
#define MF_NPH 12
#ifdef MF_STAMPS
#define MF_STAMP_DECL unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(); unsigned long long t_acc_[MF_NPH] = {};
#define MF_STAMP(ph)                                              \
    do {                                                          \
        __builtin_amdgcn_sched_barrier(0);                        \
        const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                       \
        t_acc_[ph] += t_now_ - t_prev_;                           \
        t_prev_ = t_now_;                                         \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
#define MF_STAMP_FLUSH()                                                                                         \
    do {                                                                                                         \
        if (g.stamps && lane == 0)                                                                               \
            for (int q_ = 0; q_ < MF_NPH; ++q_) g.stamps[((size_t)blockIdx.x * MF_WAVES + wave) * MF_NPH + q_] = t_acc_[q_]; \
    } while (0)
#else
#define MF_STAMP_DECL
#define MF_STAMP(ph) do { } while (0)
#define MF_STAMP_FLUSH() do { } while (0)
#endif

// ---- LDS carve (doubles) ----------------------------------------------------------------------------------
#define L_EXP 0                          // 64    exp table
#define L_PX0 64                         // 256   x0
#define L_PX1 (L_PX0 + MF_NPAD)          // 256   x1
#define L_YC (L_PX1 + MF_NPAD)           // 3*256 running right-hand sides (forward solve)
#define L_ZV (L_YC + 3 * MF_NPAD)        // 3*256 z = L^-1 y
#define L_WV (L_ZV + 3 * MF_NPAD)        // 3*256 backward-solve partial sums [8 waves][3][16] (384 used)
#define L_AV (L_WV + 3 * MF_NPAD)        // 3*256 alpha
#define L_DS (L_AV + 3 * MF_NPAD)        // 16*17 (+ pad to 288) diagonal-tile scratch
#define L_FLAG (L_DS + 288)              // 2     not-SPD flag
#define L_LINV (L_FLAG + 2)              // 16*256 L_kk^-1, operand layout
#define L_PANP (L_LINV + 16 * 256)       // 16*256 panel L_ik, operand layout (also the predict reduction buffer: 8*4*256)
#define L_TOTAL (L_PANP + 32 * 256)      // doubles

// The kernel is fully unrolled over tile slots, and every slot has its own lane-dependent LDS addresses.  hipcc
// hoists all of that loop-invariant address arithmetic out of the patch loop and keeps it in registers (hundreds
// of VGPRs, hence scratch spills).  Passing the lane id through an empty asm inside each slot body makes the
// addresses cheap-to-recompute values the compiler cannot hoist: one or two extra VALU ops per use, no spills.
__device__ static __forceinline__ int mf_opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ static inline double mf_readlane(double v, int lane_const)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_const);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_const);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to fp64 accuracy: v_rsq_f64 seed + two Newton steps
__device__ static inline double mf_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    return y;
}

// Inverse Cholesky factor of the 16 x 16 SPD tile in S (row stride 17): forward elimination on [A | I] gives
// [L^T | L^-1].  One wave; lane j < 16 holds column j of A, lane 16 + j column j of I (lanes 32..63 mirror them).
// Writes L^-1 in operand layout (element (r, c) at (r + 16 (c & 3)) * 4 + (c >> 2)) and raises *flag on a pivot <= 0.
__device__ __forceinline__ static void mf_diag_factor(const double* S, double* Linv_out, int* flag, double pivot_tol)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const bool ident = (lane & 16) != 0;
    double reg[MF_TS];
#pragma unroll
    for (int i = 0; i < MF_TS; ++i) reg[i] = ident ? (i == j ? 1.0 : 0.0) : S[i * 17 + j];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < MF_TS; ++c) {
        const double d = mf_readlane(reg[c], c);
        ok = ok && (d > pivot_tol);
        const double rs = mf_rsqrt(d);
        reg[c] *= rs;
#pragma unroll
        for (int i = c + 1; i < MF_TS; ++i) {
            const double mlt = mf_readlane(reg[i], c) * rs;
            reg[i] = __builtin_fma(-mlt, reg[c], reg[i]);
        }
    }
    if (!ok && lane == 0) *flag = 1;
    if ((lane >> 4) == 1) {
#pragma unroll
        for (int i = 0; i < MF_TS; ++i) Linv_out[(i + 16 * (j & 3)) * 4 + (j >> 2)] = reg[i];
    }
}

// ---- slot dispatch ----------------------------------------------------------------------------------------
// Tiles are enumerated column-major over the lower triangle (idx = cs(j) + i - j, cs(j) = j NT - j (j-1)/2) and
// dealt round-robin: wave w owns idx = 8 t + w in register slot t.  Every phase works on a contiguous idx range,
// i.e. on a contiguous slot range [t_lo, t_hi] of each wave, entered through a fall-through switch so that the
// scalar unit does not scan the dead slots (17 slots x 3 phases x 16 steps of compare-and-branch cost more than the
// MFMAs of the late steps).
#define MF_SLOTS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
__device__ static __forceinline__ int mf_cs(int j, int nt_full) { return j * nt_full - (j * (j - 1)) / 2; }

template <int CTRL>
__device__ static __forceinline__ double mf_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (l & 15); every lane gets the total.  row_ror:1,2,4,8 (no LDS traffic).
__device__ static __forceinline__ double mf_row_allsum(double v)
{
    v += mf_dpp<0x121>(v);
    v += mf_dpp<0x122>(v);
    v += mf_dpp<0x124>(v);
    v += mf_dpp<0x128>(v);
    return v;
}

template <int NT>
__global__ __launch_bounds__(MF_THREADS, 2) void dense_mfma_kernel(MfmaParams g)
{
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int TPW = (NTILES + MF_WAVES - 1) / MF_WAVES;
    static_assert(TPW <= 17, "MF_SLOTS covers 17 slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* T = lds + L_EXP;
    double* px0 = lds + L_PX0;
    double* px1 = lds + L_PX1;
    double* yc = lds + L_YC;
    double* zv = lds + L_ZV;
    double* wpart = lds + L_WV;
    double* av = lds + L_AV;
    double* DS = lds + L_DS;
    int* flag = reinterpret_cast<int*>(lds + L_FLAG);
    double* Linv = lds + L_LINV;
    double* panP = lds + L_PANP;

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ny = A.ny, m = A.m;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;

    // tile coordinates of this wave's slots (wave-uniform, SGPRs)
    int tij[TPW];
#define ti_(t) (tij[t] & 255)
#define tj_(t) (tij[t] >> 8)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int idx = t * MF_WAVES + wave;
        int jj = 0;
#pragma unroll
        for (int j = 1; j < NT; ++j)
            if (idx >= j * NT - (j * (j - 1)) / 2) jj = j;
        const int ii = jj + idx - (jj * NT - (jj * (jj - 1)) / 2);
        tij[t] = __builtin_amdgcn_readfirstlane((idx < NTILES) ? (ii | (jj << 8)) : (255 | (255 << 8)));
    }

    gpc_exp_table_init(T);
    MF_STAMP_DECL

    // one workgroup per patch, straight-line (a persistent patch loop makes hipcc hoist hundreds of lane-dependent
    // LDS addresses out of it and spill 1.8 KB/lane; the block hand-over costs ~1-2 us against >100 us of work)
    do {
        const int patch = blockIdx.x;
        const int o = A.off[patch];
        const int n = A.off[patch + 1] - o;
        double* fs = A.f_star + (size_t)patch * ny * m;
        if (n <= 0 || n > NT * MF_TS) {
            for (int p = tid; p < m * ny; p += MF_THREADS) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            continue;
        }
        const int nt = (n + MF_TS - 1) / MF_TS;   // live tile rows
        for (int i = tid; i < NT * MF_TS; i += MF_THREADS) {
            const bool live = i < n;
            px0[i] = live ? A.x0[o + i] : 0.0;
            px1[i] = live ? A.x1[o + i] : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < ny) yc[c * MF_NPAD + i] = live ? A.y[(size_t)c * A.n_total + o + i] : 0.0;
        }
        if (tid == 0) *flag = 0;
        __syncthreads();

        // ---- Gram tiles, transposed: acc[t][r] = K[16 i + (l&15)][16 j + (l>>4) + 4 r]; padding = identity ----
        d4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            acc[t] = d4{0.0, 0.0, 0.0, 0.0};
            if (ti_(t) < nt) {
                const int ln = mf_opaque(lane), lr = ln & 15, lg = ln >> 4;
                const int pi = MF_TS * ti_(t) + lr;
                const double xi0 = px0[pi], xi1 = px1[pi];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pj = MF_TS * tj_(t) + lg + 4 * r;
                    double v = gpc_rbf(sf, cexp, xi0, xi1, px0[pj], px1[pj], T);
                    if (pi == pj) {
                        v += noise;                              // covariance_matrix(..., training)  :59-61
                        if (A.prm.ref_double_noise) v += noise;  // C.diagonal() += sigman_sq        :21
                    }
                    if (pi >= n || pj >= n) v = (pi == pj) ? 1.0 : 0.0;
                    acc[t][r] = v;
                }
            }
        }
        MF_STAMP(0);

        // ---- right-looking tiled Cholesky, forward solve riding along ----
        // Iteration k: [B1] z_k and the TRSM of tile column k [B2] y update, trailing update with panel k.  In the
        // update the tile (k+1, k+1) goes first and its owner factors it at once (the only inlined copy of
        // mf_diag_factor), so that the serial 16 x 16 factorisation overlaps the other waves' MFMAs.  k = -1 is the
        // virtual step that only factors tile (0, 0).
        bool bad = false;
        for (int k = -1; k < nt; ++k) {
            if (k >= 0) {
                __syncthreads();   // B1: L_kk^-1 published, step k-1 updates finished (panel free to overwrite)
                MF_STAMP(1);
                if (*flag) { bad = true; break; }
                // z_k = L_kk^-1 y_k (y_k already carries -sum_{j<k} L_kj z_j): 16 row-threads of one wave
                if (wave == (k & 7) && lane < 16) {
                    const double* Lr = Linv + k * 256 + lane * 4;          // row `lane`: 4 chunks of 4 (k = g + 4 s)
                    d4 ch[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) ch[gq] = *reinterpret_cast<const d4*>(Lr + gq * 64);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (c < ny) {
                            const double* yk = yc + c * MF_NPAD + MF_TS * k;
                            double s_ = 0.0;
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                                for (int s = 0; s < 4; ++s) s_ = __builtin_fma(ch[gq][s], yk[gq + 4 * s], s_);
                            zv[c * MF_NPAD + MF_TS * k + lane] = s_;
                        }
                    }
                }
                // panel TRSM: L_ik^T = L_kk^-1 * T_ik  (A operand = L_kk^-1 from LDS, B operand = the accumulator itself)
                {
                    const int lo_ = mf_cs(k, NT) + 1 - wave, hi_ = mf_cs(k + 1, NT) - 1 - wave;
                    const int t_lo = (lo_ + 7) >> 3, t_hi = hi_ >> 3;
                    if (t_lo <= t_hi) {
                        const d4 lv = *reinterpret_cast<const d4*>(Linv + k * 256 + mf_opaque(lane) * 4);
                        switch (t_lo) {
#define MF_TRSM_CASE(t)                                                                                              \
    case t:                                                                                                          \
        if constexpr (t < TPW) {                                                                                     \
            if (t > t_hi) break;                                                                                     \
            if (ti_(t) < nt) {                                                                                       \
                d4 D1 = d4{0.0, 0.0, 0.0, 0.0}, D2 = d4{0.0, 0.0, 0.0, 0.0};                                         \
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], acc[t][0], D1, 0, 0, 0);                            \
                D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], acc[t][2], D2, 0, 0, 0);                            \
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], acc[t][1], D1, 0, 0, 0);                            \
                D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], acc[t][3], D2, 0, 0, 0);                            \
                acc[t] = D1 + D2; /* = L_ik[l & 15][(l>>4) + 4 r] */                                                 \
                *reinterpret_cast<d4*>(panP + ti_(t) * 256 + mf_opaque(lane) * 4) = acc[t];                          \
            }                                                                                                        \
        }                                                                                                            \
        [[fallthrough]];
                            MF_SLOTS(MF_TRSM_CASE)
                            default: break;
                        }
                    }
                }
                MF_STAMP(2);
                __syncthreads();   // B2: panel k and z_k complete
                MF_STAMP(3);
            }
            // pass 1: the next diagonal tile: T_(k+1)(k+1) -= L_(k+1)k L_(k+1)k^T, then straight to the factor scratch
            if (k + 1 < nt) {
                const int idx1 = mf_cs(k + 1, NT);
                if (wave == (idx1 & 7)) {
                    const int ln = mf_opaque(lane), lr = ln & 15, lg = ln >> 4;
                    switch (idx1 >> 3) {
#define MF_DIAG_CASE(t)                                                                                              \
    case t:                                                                                                          \
        if constexpr (t < TPW) {                                                                                     \
            if (k >= 0) {                                                                                            \
                const d4 a = *reinterpret_cast<const d4*>(panP + (k + 1) * 256 + ln * 4);                            \
                /* blgp = 1 on the f64 MFMA is NEG(A): acc - a*b  (tools/probe_mfma_f64.hip) */                      \
                _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                        \
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], a[s], acc[t], 0, 0, 1);                      \
            }                                                                                                        \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) DS[(lg + 4 * r) * 17 + lr] = acc[t][r];                    \
        }                                                                                                            \
        break;
                        MF_SLOTS(MF_DIAG_CASE)
                        default: break;
                    }
                    MF_STAMP(4);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    mf_diag_factor(DS, Linv + (k + 1) * 256, flag, g.pivot_tol);
                    MF_STAMP(5);
                }
            }
            if (k >= 0) {
                // forward solve: y_i -= L_ik z_k for the panel rows, one thread per matrix row, operands from LDS
                if (tid < MF_TS * (nt - 1 - k)) {
                    const int i = k + 1 + (tid >> 4), mr = tid & 15;
                    const double* Pr = panP + i * 256 + mr * 4;
                    d4 ch[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) ch[gq] = *reinterpret_cast<const d4*>(Pr + gq * 64);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (c < ny) {
                            const double* zk = zv + c * MF_NPAD + MF_TS * k;
                            double s_ = 0.0;
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                                for (int s = 0; s < 4; ++s) s_ = __builtin_fma(ch[gq][s], zk[gq + 4 * s], s_);
                            yc[c * MF_NPAD + MF_TS * i + mr] -= s_;
                        }
                    }
                }
                // pass 2: the rest of the trailing matrix, T_ij -= L_jk L_ik^T  (idx > cs(k+1))
                const int t_first = (mf_cs(k + 1, NT) + 1 - wave + 7) >> 3;
                switch (t_first) {
#define MF_UPD_CASE(t)                                                                                               \
    case t:                                                                                                          \
        if constexpr (t < TPW) {                                                                                     \
            if (ti_(t) < nt) {                                                                                       \
                const int ln4 = mf_opaque(lane) * 4;                                                                 \
                const d4 a = *reinterpret_cast<const d4*>(panP + tj_(t) * 256 + ln4);                                \
                const d4 b = *reinterpret_cast<const d4*>(panP + ti_(t) * 256 + ln4);                                \
                _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                        \
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 1);                      \
            }                                                                                                        \
        }                                                                                                            \
        [[fallthrough]];
                    MF_SLOTS(MF_UPD_CASE)
                    default: break;
                }
            }
            MF_STAMP(6);
        }
        if (bad) {
            __syncthreads();
            for (int p = tid; p < m * ny; p += MF_THREADS) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = tid; i < n * ny; i += MF_THREADS)
                    A.alpha_out[(size_t)(i / n) * A.n_total + o + (i % n)] = __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            continue;
        }
        MF_STAMP(7);

        // ---- backward solve L^T alpha = z, tile columns from the last to the first ----
        // alpha_k = L_kk^-T (z_k - sum_{i>k} L_ik^T alpha_i).  The column-k tiles sit in registers as
        // L_ik[l&15][(l>>4)+4r]: each wave sums its tiles' products in registers, reduces over the 16 lanes of a DPP
        // row (row_ror, no LDS), and publishes one 16-vector per channel; one wave finishes alpha_k.
        for (int k = nt - 1; k >= 0; --k) {
            const int lo_ = mf_cs(k, NT) + 1 - wave, hi_ = mf_cs(k + 1, NT) - 1 - wave;
            const int t_lo = (lo_ + 7) >> 3, t_hi = hi_ >> 3;
            for (int c = 0; c < ny; ++c) {      // one channel at a time keeps the register footprint at 4 doubles
                const int ln = mf_opaque(lane), lr = ln & 15, lg = ln >> 4;
                d4 pa = d4{0.0, 0.0, 0.0, 0.0};
                if (t_lo <= t_hi) {
                    const double* avc = av + c * MF_NPAD + lr;
                    switch (t_lo) {
#define MF_BWD_CASE(t)                                                                                               \
    case t:                                                                                                          \
        if constexpr (t < TPW) {                                                                                     \
            if (t > t_hi) break;                                                                                     \
            if (ti_(t) < nt) pa += acc[t] * avc[MF_TS * ti_(t)];                                                     \
        }                                                                                                            \
        [[fallthrough]];
                        MF_SLOTS(MF_BWD_CASE)
                        default: break;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) pa[s] = mf_row_allsum(pa[s]);
                }
                if (lr == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) wpart[(wave * 3 + c) * 16 + lg + 4 * s] = pa[s];
                }
            }
            __syncthreads();
            if (wave == (k & 7)) {
                // lane c' < 16: u[c'] = z_k[c'] - w_k[c'];  alpha_k[c'] = sum_m Linv[m][c'] u[m]
                const int cc = mf_opaque(lane) & 15;
                const double* Lc = Linv + k * 256 + (16 * (cc & 3)) * 4 + (cc >> 2);
                for (int c = 0; c < ny; ++c) {
                    double w_ = 0.0;
#pragma unroll
                    for (int w8 = 0; w8 < MF_WAVES; ++w8) w_ += wpart[(w8 * 3 + c) * 16 + cc];
                    const double u = zv[c * MF_NPAD + MF_TS * k + cc] - w_;
                    double a_ = 0.0;
#pragma unroll
                    for (int mm = 0; mm < MF_TS; ++mm) a_ = __builtin_fma(Lc[mm * 4], mf_readlane(u, mm), a_);
                    if (lane < 16) av[c * MF_NPAD + MF_TS * k + cc] = a_;
                }
            }
            __syncthreads();
        }
        if (A.alpha_out)
            for (int i = tid; i < n; i += MF_THREADS)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) A.alpha_out[(size_t)c * A.n_total + o + i] = av[c * MF_NPAD + i];
        MF_STAMP(8);

        // ---- predictive mean ----
        if (A.xs0 == nullptr && A.grid_sz <= 32) {
            // separable grid: f[py][px] = sum_i Ey[py][i] * (sf alpha_i Ex[px][i]); wave w takes i in [32 w, 32 w + 32)
            const int lr = lane & 15, lg = lane >> 4;
            const int sz = A.grid_sz;
            const double res = A.grid_res;
            double* red = panP;   // 8 waves x 4 tiles x 256 doubles = 64 KB: aliases the (dead) panels
            double ea[2][8], eb[2][8];
            const int ibase = 32 * wave;
            const bool wave_live = ibase < n;
            if (wave_live) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pq = 16 * h + lr;
                    const double gq = res * (((double)pq + 0.5) / (double)sz - 0.5);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ibase + 4 * s + lg;
                        const bool on = (pq < sz) && (i < n);
                        const double dy = gq - px1[i], dx = gq - px0[i];
                        ea[h][s] = on ? gpc_exp_tbl(cexp * (dy * dy), T) : 0.0;   // Ey[py = pq][i]
                        eb[h][s] = on ? gpc_exp_tbl(cexp * (dx * dx), T) : 0.0;   // Ex[px = pq][i]
                    }
                }
            }
            for (int c = 0; c < ny; ++c) {
                d4 P[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) P[mt][nl] = d4{0.0, 0.0, 0.0, 0.0};
                if (wave_live) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ibase + 4 * s + lg;
                        const double al = sf * av[c * MF_NPAD + i];
#pragma unroll
                        for (int nl = 0; nl < 2; ++nl) {
                            const double bop = eb[nl][s] * al;
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                P[mt][nl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[mt][s], bop, P[mt][nl], 0, 0, 0);
                        }
                    }
                }
                __syncthreads();   // previous channel's reduction finished reading `red` (and panels are dead)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl)
                        *reinterpret_cast<d4*>(red + ((wave * 4 + mt * 2 + nl) * 256) + lane * 4) = P[mt][nl];
                __syncthreads();
                for (int oo = tid; oo < 1024; oo += MF_THREADS) {
                    const int tile = oo >> 8, e = oo & 255, l2 = e >> 2, r = e & 3;
                    const int py = 16 * (tile >> 1) + (l2 >> 4) + 4 * r, pxx = 16 * (tile & 1) + (l2 & 15);
                    if (py < sz && pxx < sz) {
                        double s_ = 0.0;
#pragma unroll
                        for (int w = 0; w < MF_WAVES; ++w) s_ += red[(w * 4 + tile) * 256 + e];
                        fs[(size_t)c * m + py * sz + pxx] = s_;
                    }
                }
            }
        } else {
            // point-wise X* (or a grid wider than 32): one thread per prediction point
            for (int p = tid; p < m; p += MF_THREADS) {
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
                    const double kk = gpc_rbf(sf, cexp, px0[i], px1[i], q0, q1, T);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ny) s_[c] += kk * av[c * MF_NPAD + i];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) fs[(size_t)c * m + p] = s_[c];
            }
        }
        MF_STAMP(9);
        MF_STAMP_FLUSH();
        if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_OK;
    } while (0);
}

bool dense_mfma_supported(const DenseArgs& a)
{
    return a.n_max <= MF_NPAD && a.v_star == nullptr && (a.ny == 1 || a.ny == 3);
}

template <int NT>
static int launch_nt(gpc_ctx* ctx, const MfmaParams& g, int grid, const char* name)
{
    const size_t lds = sizeof(double) * (size_t)L_TOTAL;
    static bool attr_set = false;
    if (!attr_set) {
        GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_mfma_kernel<NT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(dense_mfma_kernel<NT>, dim3(grid), dim3(MF_THREADS), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    ctx->last_dense_kernel = name;
    return GPC_OK;
}

int dense_mfma_launch(gpc_ctx* ctx, const DenseArgs& a)
{
    MfmaParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.stamps = nullptr;
    const int grid = a.P;         // one workgroup per patch
#ifdef MF_STAMPS
    GPC_HIP(ctx, hipMalloc(&g.stamps, sizeof(unsigned long long) * (size_t)grid * MF_WAVES * MF_NPH));
    GPC_HIP(ctx, hipMemsetAsync(g.stamps, 0, sizeof(unsigned long long) * (size_t)grid * MF_WAVES * MF_NPH, ctx->stream));
    struct StampDump {
        gpc_ctx* ctx; unsigned long long* d; int grid;
        ~StampDump()
        {
            std::vector<unsigned long long> h((size_t)grid * MF_WAVES * MF_NPH);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            static const char* names[MF_NPH] = {"load+gram", "wait_B1", "z+trsm", "wait_B2", "pass1(diag tile upd)", "diag_factor",
                                                "yupd+pass2(update)", "post-loop", "backward", "predict", "-", "-"};
            fprintf(stderr, "[MF_STAMPS] mean cycles per patch, by wave (s_memtime ticks):\n%-22s", "phase");
            for (int w = 0; w < MF_WAVES; ++w) fprintf(stderr, "   wave%d", w);
            fprintf(stderr, "     mean\n");
            double tot = 0;
            for (int q = 0; q < 10; ++q) {
                fprintf(stderr, "%-22s", names[q]);
                double rowsum = 0;
                for (int w = 0; w < MF_WAVES; ++w) {
                    double s_ = 0;
                    for (int b = 0; b < grid; ++b) s_ += (double)h[((size_t)b * MF_WAVES + w) * MF_NPH + q];
                    s_ /= grid;
                    rowsum += s_;
                    fprintf(stderr, " %7.0f", s_);
                }
                fprintf(stderr, "  %7.0f\n", rowsum / MF_WAVES);
                tot += rowsum / MF_WAVES;
            }
            fprintf(stderr, "%-22s total %.0f ticks per patch\n", "", tot);
        }
    } dump{ctx, g.stamps, grid};
#endif
    if (a.n_max <= 64) return launch_nt<4>(ctx, g, grid, "dense_mfma_nt4");
    if (a.n_max <= 128) return launch_nt<8>(ctx, g, grid, "dense_mfma_nt8");
    if (a.n_max <= 192) return launch_nt<12>(ctx, g, grid, "dense_mfma_nt12");
    return launch_nt<16>(ctx, g, grid, "dense_mfma_nt16");
}
