// dense_mfma.hip -- register-resident batched dense GP for n <= 256 on gfx950 (the bench path, BASELINE config 2).
//
// Same computation as dense_generic.hip (gaussian_process::add_measurements + predict_measurements,
// /root/reference/src/gaussian_process.cpp:15-45), restructured for CDNA4:
//
//   * one 512-thread workgroup per patch: 7 WORKER waves + 1 FACTOR wave (wave specialisation).  The whole lower
//     triangle of K -- 136 tiles of 16 x 16 doubles = 272 KB, more than the 160 KB of LDS -- lives in the VGPR
//     file of the workers as v_mfma_f64_16x16x4_f64 accumulators (20 tile slots = 160 VGPRs per wave).  K never
//     touches HBM or L2.
//   * right-looking tiled Cholesky.  Per tile column k the panel TRSM is L_ik^T = L_kk^-1 * A_ik^T (4 MFMAs per
//     tile) and the trailing update A_ij -= L_ik L_jk^T is 4 MFMAs per tile, both operands read from a 32 KB LDS
//     panel in "operand layout" (lane l holds row l&15, k = (l>>4)+4s: one conflict-free 32-byte read per operand).
//   * tiles are stored TRANSPOSED (T_ij = A_ij^T in the MFMA C/D layout), which makes the accumulator registers of
//     a panel tile directly the B operand of its TRSM and the TRSM result directly the operand-layout image of
//     L_ik: no cross-lane movement anywhere in the O(n^3) part.
//   * the serial part -- inverting the Cholesky factor of the next 16 x 16 diagonal tile (square-root-free
//     elimination on [A | I] in registers, lane = column, wave-uniform multipliers through v_readlane) and the
//     16-wide pieces of the forward solve -- runs on the factor wave, which holds no accumulators: it overlaps the
//     workers' MFMAs instead of stalling them, and its 32-register working set does not collide with the 160
//     accumulator registers.  Workers and factor wave hand over through two LDS words (tile_ready / ready) and one
//     workgroup barrier per tile column; the panel is double-buffered so that the next TRSM may start while slow
//     workers still read the previous panel.
//   * the backward solve walks the register-resident factor (DPP row reductions, ds_add_f64), and the predictive
//     mean on the sz x sz decompression grid is evaluated separably (K* = Ex o Ey,
//     /root/reference/src/gp_compressor.cpp:317-332) as 4 more MFMA tiles per wave.
//
// Lane maps (verified by tools/probe_mfma_f64.hip): A operand lane l = A[l&15][l>>4], B operand lane l =
// B[l>>4][l&15], C/D register r of lane l = D[(l>>4) + 4r][l&15]; blgp = 1 negates A.  The whole data flow is
// replayed lane by lane in tests/test_mfma_layout_model.py.
#include <vector>

#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"


#define MF_THREADS 512
#define MF_WAVES 8
#define MF_WORKERS 7
#define MF_FACTOR_WAVE 7
#define MF_NPAD 256

struct MfmaParams {
    DenseArgs a;
    double c_exp;
    double pivot_tol;               // a pivot <= pivot_tol is reported as GPC_STATUS_NOT_SPD
    unsigned long long* stamps;     // diagnostic build (-DMF_STAMPS) only: [block][wave][MF_NPH] cycle sums
    double* export_L;               // EXPORT instantiation (predictive variance, dense_variance.hip): per patch NT (NT + 1) / 2 operand
                                    // images, row-major over the lower triangle: slot i (i + 1) / 2 + k holds L_ik (k < i) or L_ii^-1 (k == i)
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
#define MF_STAMP_FINE(ph) do { if (MF_STAMPS > 1) MF_STAMP(ph); } while (0)
#define MF_STAMP_FLUSH()                                                                                         \
    do {                                                                                                         \
        if (g.stamps && lane == 0)                                                                               \
            for (int q_ = 0; q_ < MF_NPH; ++q_) g.stamps[((size_t)blockIdx.x * MF_WAVES + wave) * MF_NPH + q_] = t_acc_[q_]; \
    } while (0)
#else
#define MF_STAMP_DECL
#define MF_STAMP(ph) do { } while (0)
#define MF_STAMP_FINE(ph) do { } while (0)
#define MF_STAMP_FLUSH() do { } while (0)
#endif

// Diagnostic timeline (-DMF_TRACE, tools/stamp_mfma.py --trace): lane 0 of every wave drops s_memtime at fixed points of
// every factorisation step into global memory; the host prints the mean timeline of factor wave and workers.
#define MF_NTR 168
#ifdef MF_TRACE
#define MF_TRACE_AT(slot)                                                                                            \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        const unsigned long long t_tr_ = __builtin_amdgcn_s_memtime();                                               \
        if (lane == 0) g.stamps[((size_t)blockIdx.x * MF_WAVES + wave) * MF_NTR + (slot)] = t_tr_;                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#else
#define MF_TRACE_AT(slot) do { } while (0)
#endif

// ---- LDS carve (doubles) ----------------------------------------------------------------------------------
// NT <= 16 (n <= 256): 256 padded points, three vector planes (depth or RGB), 16 images per array -- 158.8 KB.
// NT == 17 (n <= 272, the patches just above 256 points that an octree leaf of the C2 cloud produces): 272 padded points,
// ONE vector plane (ny == 1 only: the two colour planes are what pays for the 17th image of each array) -- 155.3 KB.
//   exp table 64 | x0, x1 NPAD each | y (running rhs), z, w (ds_add_f64 targets), alpha: NPL planes of NPAD each |
//   diagonal-tile hand-over 256 + rsqrt row (288) | flags: ints [0] not-SPD/timeout, [1] ready, [2] tile_ready, [3] alpha_ready,
//   [4] sub_ready, [5] pan_cnt, [6] t00_ready, [7] z_ready, [8 ..] pre_cnt[k] | NI images L_kk^-1 | NI images L_kk^-T |
//   2 x NI images: panel L_ik, double-buffered by k & 1 (also the predict reduction buffer: 8*4*256)
__host__ __device__ constexpr int mf_npad(int nt) { return nt > 16 ? nt * MF_TS : MF_NPAD; }
__host__ __device__ constexpr int mf_npl(int nt) { return nt > 16 ? 1 : 3; }
__host__ __device__ constexpr int mf_ni(int nt) { return nt > 16 ? nt : 16; }
__host__ __device__ constexpr int mf_flagd(int nt) { return ((4 + (mf_ni(nt) + 1) / 2) + 1) & ~1; }
__host__ __device__ constexpr int mf_l_vec(int nt) { return 64 + 2 * mf_npad(nt); }
__host__ __device__ constexpr int mf_l_ds(int nt) { return mf_l_vec(nt) + 4 * mf_npl(nt) * mf_npad(nt); }
__host__ __device__ constexpr int mf_l_flag(int nt) { return mf_l_ds(nt) + 288; }
__host__ __device__ constexpr int mf_l_linv(int nt) { return mf_l_flag(nt) + mf_flagd(nt); }
__host__ __device__ constexpr int mf_l_total(int nt) { return mf_l_linv(nt) + 4 * mf_ni(nt) * 256; }
static_assert(mf_l_total(16) == 64 + 14 * 256 + 288 + 12 + 64 * 256, "the n <= 256 carve is the one of round 1");
static_assert(mf_l_total(17) * 8 <= 160 * 1024 && (mf_l_linv(17) & 1) == 0, "NT = 17 fits the LDS, images 16-byte aligned");

// ---- slot dispatch ----------------------------------------------------------------------------------------
// Tiles are enumerated column-major over the lower triangle (idx = cs(j) + i - j, cs(j) = j NT - j (j-1)/2) and
// dealt round-robin to the 7 workers: worker w owns idx = 7 t + w in register slot t.  Every phase works on a
// contiguous idx range, i.e. on a contiguous slot range [t_lo, t_hi] of each worker; every slot body is guarded by one
// scalar range test.  (A fall-through switch into the unrolled bodies was tried: it wrecks register allocation.)
// The guards are bit tests on a wave-uniform slot mask (range & live), entered group-wise (5 slots per group) so that
// the scalar unit skips dead slots in bulk: a plain per-slot compare chain costs ~0.7k cycles per scan, and there are 64
// scans per patch.
#define MF_G0(X) X(0) X(1) X(2) X(3) X(4)
#define MF_G1(X) X(5) X(6) X(7) X(8) X(9)
#define MF_G2(X) X(10) X(11) X(12) X(13) X(14)
#define MF_G3(X) X(15) X(16) X(17) X(18) X(19)
#define MF_G4(X) X(20) X(21) X(22) X(23) X(24)
#define MF_SLOTS(X)                           \
    if (smask & 0x0001Fu) { MF_G0(X) }        \
    if (smask & 0x003E0u) { MF_G1(X) }        \
    if (smask & 0x07C00u) { MF_G2(X) }        \
    if (smask & 0xF8000u) { MF_G3(X) }        \
    if (smask & 0x1F00000u) { MF_G4(X) }
// bits t_lo .. t_hi (empty when t_hi < t_lo); 0 <= t_lo, t_hi < 31
__device__ static __forceinline__ unsigned mf_range_mask(int t_lo, int t_hi)
{
    return (t_hi >= t_lo) ? (((2u << t_hi) - 1u) & ~((1u << t_lo) - 1u)) : 0u;
}
__device__ static __forceinline__ int mf_cs(int j, int nt_full) { return j * nt_full - (j * (j - 1)) / 2; }

template <int NT, bool EXPORT = false>
__global__ __launch_bounds__(MF_THREADS, 2) void dense_mfma_kernel(MfmaParams g)
{
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int TPW = (NTILES + MF_WORKERS - 1) / MF_WORKERS;
    static_assert(TPW <= 25, "MF_G0..MF_G4 cover 25 slots");
    constexpr int NPAD = mf_npad(NT), NI = mf_ni(NT);
    constexpr int L_PX0 = 64, L_PX1 = L_PX0 + NPAD, L_YC = mf_l_vec(NT), L_ZV = L_YC + mf_npl(NT) * NPAD, L_WV = L_ZV + mf_npl(NT) * NPAD,
                  L_AV = L_WV + mf_npl(NT) * NPAD, L_DS = mf_l_ds(NT), L_FLAG = mf_l_flag(NT), L_LINV = mf_l_linv(NT),
                  L_LINVT = L_LINV + NI * 256, L_PANP = L_LINVT + NI * 256;
    constexpr int L_EXP = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* T = lds + L_EXP;
    double* px0 = lds + L_PX0;
    double* px1 = lds + L_PX1;
    double* yc = lds + L_YC;
    double* zv = lds + L_ZV;
    double* wsum = lds + L_WV;
    double* av = lds + L_AV;
    double* DS = lds + L_DS;
    int* flag = reinterpret_cast<int*>(lds + L_FLAG);   // [0] not-SPD / protocol timeout
    int* ready = flag + 1;        // highest tile column whose L_kk^-1 and z_k are published by the factor wave
    int* tile_ready = flag + 2;   // highest diagonal tile (j, j), j >= 1, handed to the factor wave (updated through panel j-2)
    int* alpha_ready = flag + 3;
    int* sub_ready = flag + 4;    // highest sub-diagonal tile (j, j-1) handed to the factor wave (updated through panel j-2)
    int* pan_cnt = flag + 5;      // panel-complete counter: 8 increments per tile column (7 workers + the factor wave)
    int* t00_ready = flag + 6;    // tile (0, 0) handed over
    int* z_ready = flag + 7;      // highest tile column whose z_k = L_kk^-1 y_k is published (initially -1)
    int* pre_cnt = flag + 8;      // one counter per tile column: tiles (i, k), i >= k+2, already added into w_k
    // 32-bit LDS byte addresses of the words for the ds_read polling loops (dynamic LDS starts after the static part)
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
    const unsigned flag_addr = lds0 + (unsigned)(L_FLAG * 8);
    const unsigned ready_addr = flag_addr + 4, tile_ready_addr = flag_addr + 8, alpha_ready_addr = flag_addr + 12;
    const unsigned sub_ready_addr = flag_addr + 16, pan_cnt_addr = flag_addr + 20, t00_ready_addr = flag_addr + 24;
    const unsigned z_ready_addr = flag_addr + 28;
    const unsigned pre_cnt_addr = flag_addr + 32;
    double* Linv = lds + L_LINV;
    double* LinvT = lds + L_LINVT;
    double* panBase = lds + L_PANP;
    // Image slot 0 of either panel buffer is never part of a panel (tile row 0 has no sub-diagonal tiles):
    double* SubX = panBase;               // hand-over of the sub-diagonal tile (j, j-1) to the factor wave (register layout)
    double* Gzero = panBase + NI * 256;   // hand-over of tile (0, 0), afterwards G_0 (backward solve)

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_factor = wave == MF_FACTOR_WAVE;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny), m = A.m;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;

    // tile coordinates of this worker's slots (wave-uniform): 4 tiles per SGPR, one byte each (ti | tj << 4); NT = 17 has tile
    // index 16, so its fields are 8 bits wide: 2 tiles per SGPR
    constexpr int TB = NT > 16 ? 16 : 8, TPS = 32 / TB, TSH = TB / 2;
    constexpr unsigned TM = (1u << TSH) - 1u;
    constexpr int TQ = (TPW + TPS - 1) / TPS;
    unsigned tq[TQ];
#define ti_(t) ((int)((tq[(t) / TPS] >> (TB * ((t) % TPS))) & TM))
#define tj_(t) ((int)((tq[(t) / TPS] >> (TB * ((t) % TPS) + TSH)) & TM))
    unsigned valid_mask = 0;   // slots that hold a tile at all
#pragma unroll
    for (int q = 0; q < TQ; ++q) tq[q] = 0;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int idx = t * MF_WORKERS + wave;
        int jj = 0;
#pragma unroll
        for (int j = 1; j < NT; ++j)
            if (idx >= j * NT - (j * (j - 1)) / 2) jj = j;
        const int ii = jj + idx - (jj * NT - (jj * (jj - 1)) / 2);
        const bool ok = idx < NTILES && !is_factor;
        tq[t / TPS] |= ok ? ((unsigned)(ii | (jj << TSH)) << (TB * (t % TPS))) : 0u;
        valid_mask |= ok ? (1u << t) : 0u;
    }
#pragma unroll
    for (int q = 0; q < TQ; ++q) tq[q] = __builtin_amdgcn_readfirstlane(tq[q]);
    valid_mask = __builtin_amdgcn_readfirstlane(valid_mask);

    gpc_exp_table_init(T);
    MF_STAMP_DECL

    // one workgroup per patch, straight-line (a persistent patch loop makes hipcc hoist hundreds of lane-dependent
    // LDS addresses out of it and spill; the block hand-over costs ~1-2 us against ~100 us of work)
    // (The size-class dispatch launches P workgroups per class although only the class's share holds a patch: the count lives
    // on the device.  The empty ones each wait for a CU with 158 KB of LDS free, ~0.1 ms per launch on the C2-size cloud; walking
    // the class list with one workgroup per CU instead turns this body into a loop and costs 1.6 KB of scratch per lane --
    // measured at compile time for every instantiation, even with the loop compiled out by a template flag -- so it stays.)
    do {
        int patch = blockIdx.x;
        if (A.sel) {                                  // size-class dispatch: this launch owns the patches sel[0 .. count)
            if ((int)blockIdx.x >= __builtin_amdgcn_readfirstlane(A.sel_count[0])) continue;
            patch = __builtin_amdgcn_readfirstlane(A.sel[blockIdx.x]);
        }
        // wave-uniform by construction, but loaded through the vector memory path: without readfirstlane hipcc treats n
        // (and every `ti < nt` test derived from it) as divergent and lowers the slot guards to EXEC-masked code
        const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
        const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
        double* fs = A.f_star + (size_t)patch * ny * m;
        [[maybe_unused]] double* gexp = EXPORT ? g.export_L + (size_t)patch * (NT * (NT + 1) / 2) * MF_IMG : nullptr;
        if (n <= 0 || n > NT * MF_TS) {
            for (int p = tid; p < m * ny; p += MF_THREADS) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            continue;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);   // live tile rows
        // extent of the patch around its first point (max-norm): bounds every kernel argument of this patch, see below
        const double xo0 = A.x0[o], xo1 = A.x1[o];
        double dev = 0.0;
        for (int i = tid; i < NT * MF_TS; i += MF_THREADS) {
            const bool live = i < n;
            const double q0 = live ? A.x0[o + i] : xo0, q1 = live ? A.x1[o + i] : xo1;
            dev = __builtin_fmax(dev, __builtin_fmax(__builtin_fabs(q0 - xo0), __builtin_fabs(q1 - xo1)));
            px0[i] = live ? q0 : 0.0;
            px1[i] = live ? q1 : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < ny) {
                    yc[c * NPAD + i] = live ? A.y[(size_t)c * A.n_total + o + i] : 0.0;
                    wsum[c * NPAD + i] = 0.0;
                }
            }
        }
        if (tid == 0) {
            flag[0] = 0;
            flag[1] = -1;
            flag[2] = 0;
            flag[3] = 0;
            flag[4] = 0;
            flag[5] = 0;
            flag[6] = 0;
            flag[7] = -1;
        }
        if (tid < NI) flag[8 + tid] = 0;
        dev = __builtin_fmax(dev, mf_dpp<0x121>(dev));     // max over the 16 lanes of a DPP row
        dev = __builtin_fmax(dev, mf_dpp<0x122>(dev));
        dev = __builtin_fmax(dev, mf_dpp<0x124>(dev));
        dev = __builtin_fmax(dev, mf_dpp<0x128>(dev));
        if ((lane & 15) == 0) DS[tid >> 4] = dev;          // 32 row maxima (DS is free until the first tile hand-over)
        __syncthreads();
        // Small-argument regime: every Gram argument satisfies |c| d^2 <= |c| 2 (2 r)^2 and every argument of the separable
        // grid factors |c| (res/2 + |x_first| + r)^2, r = patch extent.  When both are <= 2^-5 the exponential is the plain
        // degree-7 polynomial (gpc_exp_small); otherwise the table-driven gpc_exp_neg.  Wave-uniform.
        bool small_gram, small_grid;
        {
            double r = DS[lane & 31];
            r = __builtin_fmax(r, mf_dpp<0x121>(r));
            r = __builtin_fmax(r, mf_dpp<0x122>(r));
            r = __builtin_fmax(r, mf_dpp<0x124>(r));
            r = __builtin_fmax(r, mf_dpp<0x128>(r));
            r = __builtin_fmax(mf_readlane(r, 0), mf_readlane(r, 16));
            const double b = 0.5 * A.grid_res + __builtin_fmax(__builtin_fabs(xo0), __builtin_fabs(xo1)) + r;
            small_gram = -cexp * (8.0 * r * r) <= GPC_EXP_SMALL_MAX;      // false for NaN / inf extents
            small_grid = -cexp * (b * b) <= GPC_EXP_SMALL_MAX;
        }
        __syncthreads();                                   // DS is read; the hand-over may overwrite it

        unsigned live_mask = 0;   // slots whose tile row is live (ti < nt)
#pragma unroll
        for (int t = 0; t < TPW; ++t) live_mask |= (ti_(t) < nt) ? (1u << t) : 0u;
        live_mask = __builtin_amdgcn_readfirstlane(live_mask & valid_mask);
        d4 acc[TPW];   // worker waves only; never live on the factor wave's path through the factorisation
        bool timed_out = false;
        if (!is_factor) {
            // ================================ WORKER ROLE ================================
            // ---- Gram tiles, transposed: acc[t][r] = K[16 i + (l&15)][16 j + (l>>4) + 4 r]; padding = identity ----
            // Off-diagonal interior tiles (120 of 136 at n = 256) take the bare kernel evaluation; the noise diagonal and
            // the identity padding live in a separate branch.  (Written as one body with wave-uniform conditions, hipcc
            // if-converts them into ~18 selects and compares per ELEMENT -- more than the exponential itself.  The opaque
            // lane id inside the special branch is what keeps it a branch.)
#define MF_GRAM(RBF, T0, T1)                                                                                         \
    _Pragma("unroll") for (int t = T0; t < T1; ++t) {                                                                \
        acc[t] = d4{0.0, 0.0, 0.0, 0.0};                                                                             \
        if (live_mask & (1u << t)) {                                                                                 \
            const int ln = mf_opaque(lane), lr = ln & 15, lg = ln >> 4;                                              \
            const int pi = MF_TS * ti_(t) + lr;                                                                      \
            const double xi0 = px0[pi], xi1 = px1[pi];                                                               \
            const bool edge = MF_TS * ti_(t) + MF_TS > n;   /* tile touches the identity padding (wave-uniform) */    \
            const bool diag_tile = ti_(t) == tj_(t);        /* only these carry the noise diagonal (wave-uniform) */  \
            if (__builtin_expect(edge || diag_tile, 0)) {                                                            \
                const int lg2 = mf_opaque(ln) >> 4;                                                                  \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                      \
                    const int pj = MF_TS * tj_(t) + lg2 + 4 * r;                                                     \
                    double v = RBF;                                                                                  \
                    if (diag_tile && pi == pj) {                                                                     \
                        v += noise;                              /* covariance_matrix(..., training)  :59-61 */      \
                        if (A.prm.ref_double_noise) v += noise;  /* C.diagonal() += sigman_sq        :21 */          \
                    }                                                                                                \
                    if (edge && (pi >= n || pj >= n)) v = (pi == pj) ? 1.0 : 0.0;                                    \
                    acc[t][r] = v;                                                                                   \
                }                                                                                                    \
            } else {                                                                                                 \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                      \
                    const int pj = MF_TS * tj_(t) + lg + 4 * r;                                                      \
                    acc[t][r] = RBF;                                                                                 \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }
            // The first three slots hold the tiles the factor wave starts from -- (0,0), (1,0), (1,1) for every NT -- so they are
            // evaluated and handed over FIRST: the diagonal factor of column 0, the TRSM of (1,0) and the factor of column 1 then
            // run on the factor wave while the workers are still evaluating the rest of the Gram matrix.
            constexpr int TH = TPW < 3 ? TPW : 3;
            // sigma_f^2 folded into the polynomial coefficients: sf exp(x) in 7 FMAs
            const double k7 = sf * (1.0 / 5040.0), k6 = sf * (1.0 / 720.0), k5 = sf * (1.0 / 120.0), k4 = sf * (1.0 / 24.0),
                         k3 = sf * (1.0 / 6.0), k2 = sf * 0.5;
#define MF_RBF_SMALL(xa, ya, xb, yb)                                                                                 \
    ([&]() __attribute__((always_inline)) {                                                                          \
        const double d0_ = (xa) - (xb), d1_ = (ya) - (yb);                                                           \
        const double x_ = cexp * __builtin_fma(d0_, d0_, d1_ * d1_);                                                 \
        double p_ = __builtin_fma(x_, k7, k6);                                                                       \
        p_ = __builtin_fma(x_, p_, k5);                                                                              \
        p_ = __builtin_fma(x_, p_, k4);                                                                              \
        p_ = __builtin_fma(x_, p_, k3);                                                                              \
        p_ = __builtin_fma(x_, p_, k2);                                                                              \
        p_ = __builtin_fma(x_, p_, sf);                                                                              \
        return __builtin_fma(x_, p_, sf);                                                                            \
    }())
            if (small_gram) {
                MF_GRAM(MF_RBF_SMALL(xi0, xi1, px0[pj], px1[pj]), 0, TH)
            } else {
                MF_GRAM(gpc_rbf_neg(sf, cexp, xi0, xi1, px0[pj], px1[pj], T), 0, TH)
            }
            // ---- first hand-overs: tile (0,0), and tiles (1,0) / (1,1) as they are (no panel precedes them) ----
#define MF_HAND_CASE(t)                                                                                              \
    if constexpr (t < TPW) {                                                                                         \
        if (smask & (1u << t)) *reinterpret_cast<d4*>(hand_to + mf_opaque(lane) * 4) = acc[t];                       \
    }
            static_assert(1 / MF_WORKERS == 0 && NT / MF_WORKERS < 3, "the chain's first tiles sit in slots 0 .. 2");
            if (wave == 0) {
                *reinterpret_cast<d4*>(Gzero + mf_opaque(lane) * 4) = acc[0];   // register layout, as the factor wave consumes it
                mf_publish(t00_ready, 1);
            }
            if (nt > 1) {
                if (wave == 1 % MF_WORKERS) {                       // idx 1 = tile (1, 0)
                    double* hand_to = SubX;
                    const unsigned smask = 1u << (1 / MF_WORKERS);
                    MF_SLOTS(MF_HAND_CASE)
                    mf_publish(sub_ready, 1);
                }
                if (wave == NT % MF_WORKERS) {                      // idx cs(1) = NT = tile (1, 1)
                    double* hand_to = DS;
                    const unsigned smask = __builtin_amdgcn_readfirstlane(1u << (NT / MF_WORKERS));
                    MF_SLOTS(MF_HAND_CASE)
                    mf_publish(tile_ready, 1);
                }
            }
            if (small_gram) {
                MF_GRAM(MF_RBF_SMALL(xi0, xi1, px0[pj], px1[pj]), TH, TPW)
            } else {
                MF_GRAM(gpc_rbf_neg(sf, cexp, xi0, xi1, px0[pj], px1[pj], T), TH, TPW)
            }
            MF_STAMP(0);

            // ---- right-looking tiled Cholesky, worker side ----
            // The factor wave owns the whole critical chain of a tile column: diag factor (k,k) -> TRSM of the sub-diagonal
            // tile (k+1,k) -> update and factor of (k+1,k+1).  It gets both tiles one step AHEAD (updated through panel k-1)
            // from their owners, who update and hand them over first thing after panel k-1 is complete.
            // A worker's step k starts when panel k is complete (pan_cnt == 8 (k+1): 7 workers + the factor wave; the
            // factor wave only counts up, it never waits for the workers, and there is no s_barrier in the loop):
            //   owner of (k+1,k): take the final tile back from the panel, build G_k | owners of (k+2,k+1), (k+2,k+2):
            //   update with panel k, hand over | forward-solve rows | trailing update of the columns >= k+2 | LOOK-AHEAD:
            //   its tiles of column k+1 are updated with panel k and solved with L_(k+1)(k+1)^-1 straight away -> panel k+1.
            // The TRSMs of panel k+1 thus run inside update k, interleaved with the SIMD partner's MFMAs, instead of in a
            // phase of their own between two synchronisation points.
#define MF_TRSM_CASE(t)                                                                                              \
    if constexpr (t < TPW) {                                                                                         \
        if (smask & (1u << t)) {                                                                                     \
            /* four independent products (one MFMA latency instead of four), then a tree sum */                     \
            const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};                                                                    \
            const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], acc[t][0], z4, 0, 0, 0);                       \
            const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], acc[t][1], z4, 0, 0, 0);                       \
            const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], acc[t][2], z4, 0, 0, 0);                       \
            const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], acc[t][3], z4, 0, 0, 0);                       \
            acc[t] = (D0 + D1) + (D2 + D3); /* = L_ik[l & 15][(l>>4) + 4 r] */                                       \
            mf_img_store(panN + ti_(t) * MF_IMG, mf_opaque(lane), acc[t]);                                          \
            if constexpr (EXPORT)                                                                                    \
                mf_img_store(gexp + (size_t)((ti_(t) * (ti_(t) + 1)) / 2 + tj_(t)) * MF_IMG, mf_opaque(lane), acc[t]); \
        }                                                                                                            \
    }
#define MF_UPD_CASE(t)                                                                                               \
    if constexpr (t < TPW) {                                                                                         \
        if (smask & (1u << t)) {                                                                                     \
            const int lnq = mf_opaque(lane);                                                                         \
            const d4 a = mf_img_load(panP + tj_(t) * MF_IMG, lnq);                                                   \
            const d4 b = mf_img_load(panP + ti_(t) * MF_IMG, lnq);                                                   \
            /* blgp = 1 on the f64 MFMA is NEG(A): acc - a*b  (tools/probe_mfma_f64.hip) */                          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                            \
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 1);                          \
        }                                                                                                            \
    }
#define MF_UPD_TRSM_CASE(t) MF_UPD_CASE(t) MF_TRSM_CASE(t)
#define MF_UPD_HAND_CASE(t)                                                                                          \
    if constexpr (t < TPW) {                                                                                         \
        if (smask & (1u << t)) {                                                                                     \
            const int lnq = mf_opaque(lane);                                                                         \
            const d4 a = mf_img_load(panP + tj_(t) * MF_IMG, lnq);                                                   \
            const d4 b = mf_img_load(panP + ti_(t) * MF_IMG, lnq);                                                   \
            const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};   /* independent products: one MFMA latency on the chain */       \
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[t], 0, 0, 1);                              \
            const d4 S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], z4, 0, 0, 1);                             \
            const d4 S2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], z4, 0, 0, 1);                             \
            const d4 S3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], z4, 0, 0, 1);                             \
            acc[t] += (S1 + S2) + S3;                                                                                \
            *reinterpret_cast<d4*>(hand_to + lnq * 4) = acc[t];                                                      \
        }                                                                                                            \
    }
#define MF_RELOAD_CASE(t)                                                                                            \
    if constexpr (t < TPW) {                                                                                         \
        if (smask & (1u << t)) acc[t] = Ln;                                                                          \
    }
            bool dead = false;
            // prologue: panel 0 = TRSM of column 0 with L_00^-1
            timed_out |= !mf_wait_ge(ready_addr, 0);
            dead = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            if (!dead && nt > 1) {
                double* panN = panBase;
                const int lo_ = mf_cs(0, NT) + 2 - wave, hi_ = mf_cs(1, NT) - 1 - wave;
                const int t_lo = __builtin_amdgcn_readfirstlane((lo_ + 6) / 7);
                const int t_hi = __builtin_amdgcn_readfirstlane((hi_ + 7) / 7 - 1);
                const unsigned smask = __builtin_amdgcn_readfirstlane(mf_range_mask(t_lo, t_hi) & live_mask);
                if (smask) {
                    const d4 lv = mf_img_load(Linv, mf_opaque(lane));
                    MF_SLOTS(MF_TRSM_CASE)
                }
                if (lane == 0) __hip_atomic_fetch_add(pan_cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            for (int k = 0; k + 1 < nt && !dead; ++k) {
                double* panP = panBase + (k & 1) * (NI * 256);
                double* panN = panBase + ((k + 1) & 1) * (NI * 256);
                // panel k complete?  (also: every wave has finished step k-1, so panel buffer (k+1)&1 may be rewritten)
                timed_out |= !mf_wait_ge(pan_cnt_addr, MF_WAVES * (k + 1));
                MF_STAMP_FINE(1);
                MF_TRACE_AT(8 * k + 0);
                // the owner of (k+1, k) takes the final L_(k+1)k back into its register slot (backward solve) and builds
                // G_k = L_kk^-T L_(k+1)k^T: (L_(k+1)k L_kk^-1) in C/D layout is the operand image of its transpose.  It lands
                // in the image slot of L_(k-1)(k-1)^-1, dead since panel k-1 was complete (k = 0: the tile (0,0) buffer).
                {
                    const int idxs = __builtin_amdgcn_readfirstlane(mf_cs(k, NT) + 1);
                    if (wave == idxs % MF_WORKERS) {
                        const unsigned smask = __builtin_amdgcn_readfirstlane(1u << (idxs / MF_WORKERS));
                        const int ln = mf_opaque(lane);
                        const d4 Ln = mf_img_load(panP + (k + 1) * MF_IMG, ln);
                        const d4 lt = mf_img_load(LinvT + k * MF_IMG, ln);
                        MF_SLOTS(MF_RELOAD_CASE)
                        const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                        const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[0], lt[0], z4, 0, 0, 0);
                        const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[1], lt[1], z4, 0, 0, 0);
                        const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[2], lt[2], z4, 0, 0, 0);
                        const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[3], lt[3], z4, 0, 0, 0);
                        mf_img_store(k > 0 ? Linv + (k - 1) * MF_IMG : Gzero, ln, (D0 + D1) + (D2 + D3));
                    }
                }
                // next step's chain tiles first: (k+2,k+1) -= L_(k+2)k L_(k+1)k^T and (k+2,k+2) -= L_(k+2)k L_(k+2)k^T,
                // then straight to the factor wave.
                if (k + 2 < nt) {
                    const int idxs = __builtin_amdgcn_readfirstlane(mf_cs(k + 1, NT) + 1);
                    if (wave == idxs % MF_WORKERS) {
                        const unsigned smask = __builtin_amdgcn_readfirstlane(1u << (idxs / MF_WORKERS));
                        double* hand_to = SubX;
                        MF_SLOTS(MF_UPD_HAND_CASE)
                        // this owner also applies panel k to the forward-solve rows of block k+2 (the release of sub_ready
                        // orders it before the factor wave's own update of that block)
                        timed_out |= !mf_wait_ge(z_ready_addr, k);
                        if (lane < 16)
                            for (int c = 0; c < ny; ++c)
                                yc[c * NPAD + MF_TS * (k + 2) + lane] -=
                                    mf_row_dot(panP + (k + 2) * MF_IMG, lane, zv + c * NPAD + MF_TS * k);
                        mf_publish(sub_ready, k + 2);
                    }
                    const int idxd = __builtin_amdgcn_readfirstlane(mf_cs(k + 2, NT));
                    if (wave == idxd % MF_WORKERS) {
                        const unsigned smask = __builtin_amdgcn_readfirstlane(1u << (idxd / MF_WORKERS));
                        double* hand_to = DS;
                        MF_SLOTS(MF_UPD_HAND_CASE)
                        mf_publish(tile_ready, k + 2);
                    }
                }
                MF_STAMP_FINE(4);
                MF_TRACE_AT(8 * k + 1);
                // forward-solve rows of the blocks nobody on the chain needs yet: y_i -= L_ik z_k, i >= k+3, one thread per row.
                // On the low waves only: their SIMD partners (waves 4..6) stream trailing-update MFMAs meanwhile, so in the
                // MFMA-bound early steps this work is hidden.  (Doing it from the registers right after the TRSM, before the
                // synchronisation point, was measured 12k cycles slower per patch.)
                if (tid < MF_TS * (nt - 3 - k)) {
                    timed_out |= !mf_wait_ge(z_ready_addr, k);
                    const int i = k + 3 + (tid >> 4), mr = tid & 15;
                    for (int c = 0; c < ny; ++c)
                        yc[c * NPAD + MF_TS * i + mr] -= mf_row_dot(panP + i * 256, mr, zv + c * NPAD + MF_TS * k);
                }
                MF_STAMP_FINE(5);
                MF_TRACE_AT(8 * k + 2);
                // trailing update of the columns >= k+2, T_ij -= L_jk L_ik^T: idx >= cs(k+2) + 1 ((k+2,k+2) went to the factor wave)
                {
                    const int t_first = (mf_cs(k + 2, NT) + 1 - wave + 6) / 7;
                    const unsigned smask = __builtin_amdgcn_readfirstlane(mf_range_mask(t_first, 30) & live_mask);
                    MF_SLOTS(MF_UPD_CASE)
                }
                MF_STAMP_FINE(6);
                MF_TRACE_AT(8 * k + 3);
                // look-ahead: column k+1 -- update with panel k, TRSM with L_(k+1)(k+1)^-1, store into panel k+1.
                // Every worker polls `ready` here, with or without tiles: that is where a not-SPD verdict reaches it.
                timed_out |= !mf_wait_ge(ready_addr, k + 1);
                MF_TRACE_AT(8 * k + 4);
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                if (k + 2 < nt) {
                    const int lo_ = mf_cs(k + 1, NT) + 2 - wave, hi_ = mf_cs(k + 2, NT) - 1 - wave;
                    const int t_lo = __builtin_amdgcn_readfirstlane((lo_ + 6) / 7);
                    const int t_hi = __builtin_amdgcn_readfirstlane((hi_ + 7) / 7 - 1);
                    const unsigned smask = __builtin_amdgcn_readfirstlane(mf_range_mask(t_lo, t_hi) & live_mask);
                    if (smask) {
                        const d4 lv = mf_img_load(Linv + (k + 1) * MF_IMG, mf_opaque(lane));
                        MF_SLOTS(MF_UPD_TRSM_CASE)
                    }
                    if (lane == 0) __hip_atomic_fetch_add(pan_cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                MF_TRACE_AT(8 * k + 5);
            }
        } else {
            // ================================ FACTOR ROLE ================================
            // The critical chain of the factorisation, inside one wave: for j = 0 .. nt-1
            //   W = tile (j,j)  ->  L_jj^-1, L_jj^-T (15 rank-1 MFMAs)  ->  z_j  ->  publish ready = j
            //   tiles (j+1,j), (j+1,j+1) arrive from their owners (updated through panel j-1)
            //   L_(j+1)j = TRSM (4 MFMAs) -> panel buffer, pan_cnt++  ->  (j+1,j+1) -= L L^T (4 MFMAs) = next W
            //   y_(j+1) -= L_(j+1)j z_j  in the shadow of those MFMAs.
            // It shares its SIMD (and that SIMD's FP64 pipe) with a worker that streams MFMAs: static priority lets its
            // short dependent chain win the arbitration (MI355X_MICROARCH.md, 'Two waves per SIMD', item 4).
            __builtin_amdgcn_s_setprio(3);
            MF_STAMP(0);
            timed_out |= !mf_wait_ge(t00_ready_addr, 1);
            d4 W = *reinterpret_cast<const d4*>(Gzero + mf_opaque(lane) * 4);
            for (int j = 0; j < nt; ++j) {
                MF_STAMP_FINE(4);
                MF_TRACE_AT(8 * j + 0);
                const bool ok = mf_diag_factor(W, DS + 256, Linv + j * 256, LinvT + j * 256, g.pivot_tol);
                MF_STAMP_FINE(5);
                MF_TRACE_AT(8 * j + 1);
                if (!ok && lane == 0) flag[0] = 1;
                mf_publish(ready, j);            // L_jj^-1 is all the workers need to start their TRSMs; z_j follows
                MF_TRACE_AT(8 * j + 2);
                const int ln = mf_opaque(lane);
                double* panP = panBase + (j & 1) * (NI * 256);
                const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                d4 D0 = z4, D1 = z4, D2 = z4, D3 = z4;
                const d4 lv = mf_img_load(Linv + j * MF_IMG, ln);
                if constexpr (EXPORT) mf_img_store(gexp + (size_t)((j * (j + 1)) / 2 + j) * MF_IMG, ln, lv);
                // right-hand sides of block j as MFMA B operand: column n < ny carries channel n
                d4 yb = z4;
                {
                    const int lr = ln & 15, lg = ln >> 4;
                    if (lr < ny) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) yb[q4] = yc[lr * NPAD + MF_TS * j + lg + 4 * q4];
                    }
                }
                if (ok && j + 1 < nt) {
                    // TRSM of the sub-diagonal tile: L_(j+1)j^T = L_jj^-1 A^T
                    timed_out |= !mf_wait_ge(sub_ready_addr, j + 1);
                    MF_TRACE_AT(8 * j + 3);
                    const d4 B = *reinterpret_cast<const d4*>(SubX + ln * 4);
                    D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], B[0], z4, 0, 0, 0);
                    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], B[1], z4, 0, 0, 0);
                    D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], B[2], z4, 0, 0, 0);
                    D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], B[3], z4, 0, 0, 0);
                }
                // behind them on the pipe: z_j = L_jj^-1 y_j (y_j already carries -sum_{i<j} L_ji z_i) as one more 16x16x16 product
                const d4 Z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[0], yb[0], z4, 0, 0, 0);
                const d4 Z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[1], yb[1], z4, 0, 0, 0);
                const d4 Z2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[2], yb[2], z4, 0, 0, 0);
                const d4 Z3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[3], yb[3], z4, 0, 0, 0);
                d4 Ln = z4, S1 = z4, S2 = z4, S3 = z4;
                if (ok && j + 1 < nt) {
                    Ln = (D0 + D1) + (D2 + D3);                       // operand image of L_(j+1)j
                    mf_img_store(panP + (j + 1) * MF_IMG, ln, Ln);
                    if constexpr (EXPORT) mf_img_store(gexp + (size_t)(((j + 1) * (j + 2)) / 2 + j) * MF_IMG, ln, Ln);
                    timed_out |= !mf_wait_ge(tile_ready_addr, j + 1);
                    MF_TRACE_AT(8 * j + 4);
                    W = *reinterpret_cast<const d4*>(DS + ln * 4);
                    // (j+1,j+1) -= L_(j+1)j L_(j+1)j^T: independent products, one MFMA latency on the chain
                    W = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[0], Ln[0], W, 0, 0, 1);      // blgp = 1: NEG(A)
                    S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[1], Ln[1], z4, 0, 0, 1);
                    S2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[2], Ln[2], z4, 0, 0, 1);
                    S3 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[3], Ln[3], z4, 0, 0, 1);
                }
                // in the shadow of those MFMAs: publish z_j, count up the panel, y_(j+1) -= L_(j+1)j z_j from the registers
                {
                    const d4 zj = (Z0 + Z1) + (Z2 + Z3);   // lanes lr = n < ny: z_n[16 j + (l>>4) + 4 r]
                    const int lr = ln & 15, lg = ln >> 4;
                    if (lr < ny) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) zv[lr * NPAD + MF_TS * j + lg + 4 * r] = zj[r];
                    }
                }
                mf_publish(z_ready, j);
                MF_STAMP_FINE(2);
                if (!ok) break;                                       // the workers leave at step j as well
                if (j + 1 < nt) {
                    if (lane == 0) __hip_atomic_fetch_add(pan_cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    MF_STAMP_FINE(3);
                    {
                        const int lr = ln & 15, lg = ln >> 4;
                        for (int c = 0; c < ny; ++c) {
                            const double* zq = zv + c * NPAD + MF_TS * j + lg;
                            atomicAdd(yc + c * NPAD + MF_TS * (j + 1) + lr,
                                      -((Ln[0] * zq[0] + Ln[1] * zq[4]) + (Ln[2] * zq[8] + Ln[3] * zq[12])));
                        }
                    }
                    W += (S1 + S2) + S3;
                    MF_TRACE_AT(8 * j + 5);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const bool bad = flag[0] != 0;
        if (bad) {
            for (int p = tid; p < m * ny; p += MF_THREADS) fs[p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = tid; i < n * ny; i += MF_THREADS)
                    A.alpha_out[(size_t)(i / n) * A.n_total + o + (i % n)] = __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            continue;
        }
        MF_STAMP(7);

        // ---- backward solve L^T alpha = z, tile columns from the last to the first ----
        // alpha_k = L_kk^-T (z_k - w_k) - G_k alpha_(k+1),  w_k = sum_{i>=k+2} L_ik^T alpha_i,  G_k = L_kk^-T L_(k+1)k^T.
        // The nearest tile, (k+1, k), never enters w_k: the factor wave folded it into G_k while it had both factors at
        // hand, so the alpha_(k+1) -> alpha_k dependency is 4 MFMAs inside one wave, and what the workers add needs
        // alpha_(k+2) and older only.  No workgroup barrier.
        //   factor : for k = nt-1 .. 0: waits until w_k is complete (pre_cnt[k] == number of live tiles (i, k), i >= k+2),
        //            computes alpha_k, publishes alpha_ready = nt - k;
        //   workers: ONE static sweep over their register slots from the last tile to the first.  Column-major dealing
        //            makes that sweep visit the columns in exactly the order the factor wave needs them and, inside a
        //            column, the rows from the bottom up, i.e. in the order their alpha_i appear.  Products of one
        //            column are summed in registers, reduced over the 16 lanes of a DPP row and added to w_k with
        //            ds_add_f64; pre_cnt[k] counts the tiles added.  (The previous version ran a 16-iteration loop in
        //            every worker, each iteration re-deriving its slot range and scanning 20 slot guards: 2.2k cycles per
        //            iteration of pure overhead.)
        if (is_factor) {
            d4 al = d4{0.0, 0.0, 0.0, 0.0};   // alpha of the previous iteration, B-operand layout
            for (int k = nt - 1; k >= 0; --k) {
                const int ln = mf_opaque(lane), lr = ln & 15, lg = ln >> 4;
                // operands that do not depend on the workers: fetched before the wait
                const d4 lt = mf_img_load(LinvT + k * MF_IMG, ln);
                d4 gk = d4{0.0, 0.0, 0.0, 0.0};
                if (k + 1 < nt) gk = mf_img_load(k > 0 ? Linv + (k - 1) * MF_IMG : Gzero, ln);
                // -G_k alpha_(k+1) does not depend on the workers: issued before the wait.  16x16x16 MFMA products, column
                // n < ny of the B operand carries channel n; alpha_(k+1) is still in this wave's registers from the previous
                // iteration, in exactly the B operand layout.
                const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(gk[0], al[0], z4, 0, 0, 1);   // blgp = 1: NEG(A)
                d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(gk[1], al[1], z4, 0, 0, 1);
                d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(gk[2], al[2], z4, 0, 0, 1);
                d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(gk[3], al[3], z4, 0, 0, 1);
                d4 ub = d4{0.0, 0.0, 0.0, 0.0};
                if (lr < ny) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) ub[q4] = zv[lr * NPAD + MF_TS * k + lg + 4 * q4];
                }
                const int expect = nt - k - 2;
                if (expect > 0) {
                    timed_out |= !mf_wait_ge(pre_cnt_addr + 4u * (unsigned)k, expect);
                    MF_STAMP_FINE(10);
                    if (lr < ny) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) ub[q4] -= wsum[lr * NPAD + MF_TS * k + lg + 4 * q4];
                    }
                }
                D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[0], ub[0], D0, 0, 0, 0);       // + L_kk^-T (z_k - w_k)
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[1], ub[1], D1, 0, 0, 0);
                D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[2], ub[2], D2, 0, 0, 0);
                D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[3], ub[3], D3, 0, 0, 0);
                al = (D0 + D1) + (D2 + D3);   // lanes lr = n < ny: alpha_n[16 k + (l>>4) + 4 r]; zero elsewhere
                if (lr < ny) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[lr * NPAD + MF_TS * k + lg + 4 * r] = al[r];
                }
                mf_publish(alpha_ready, nt - k);
                MF_STAMP_FINE(11);
            }
        } else {
            int cur_col = -1, cnt = 0, known = 0;
            d4 pa[3];
#define MF_BWD_FLUSH()                                                                                               \
    do {                                                                                                             \
        const int lnf = mf_opaque(lane), lrf = lnf & 15, lgf = lnf >> 4;                                             \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {   /* static index: pa[] must stay in registers */            \
            if (c < ny) {                                                                                            \
                const double tot = mf_row_reduce4(pa[c], lrf);   /* lanes lr = 0, 4, 8, 12 hold components 0..3 */   \
                if ((lrf & 3) == 0) atomicAdd(wsum + c * NPAD + MF_TS * cur_col + lgf + 4 * (lrf >> 2), tot);     \
            }                                                                                                        \
        }                                                                                                            \
        if (lane == 0) __hip_atomic_fetch_add(pre_cnt + cur_col, cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)
#define MF_BWD_CASE(t)                                                                                               \
    if constexpr (t < TPW) {                                                                                         \
        if ((live_mask & (1u << t)) && ti_(t) >= tj_(t) + 2) {                                                       \
            if (tj_(t) != cur_col) {                                                                                 \
                if (cnt) MF_BWD_FLUSH();                                                                             \
                cur_col = tj_(t);                                                                                    \
                cnt = 0;                                                                                             \
                _Pragma("unroll") for (int c = 0; c < 3; ++c) pa[c] = d4{0.0, 0.0, 0.0, 0.0};                        \
            }                                                                                                        \
            const int need = nt - ti_(t);   /* alpha_ti is published as alpha_ready = nt - ti */                     \
            if (known < need) {                                                                                      \
                timed_out |= !mf_wait_ge(alpha_ready_addr, need);                                                    \
                known = need;                                                                                        \
            }                                                                                                        \
            const double* avq = av + MF_TS * ti_(t) + (mf_opaque(lane) & 15);                                        \
            _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                            \
                if (c < ny) pa[c] += acc[t] * avq[c * NPAD];                                                      \
            ++cnt;                                                                                                   \
        }                                                                                                            \
    }
            MF_BWD_CASE(24) MF_BWD_CASE(23) MF_BWD_CASE(22) MF_BWD_CASE(21) MF_BWD_CASE(20)
            MF_BWD_CASE(19) MF_BWD_CASE(18) MF_BWD_CASE(17) MF_BWD_CASE(16) MF_BWD_CASE(15)
            MF_BWD_CASE(14) MF_BWD_CASE(13) MF_BWD_CASE(12) MF_BWD_CASE(11) MF_BWD_CASE(10)
            MF_BWD_CASE(9) MF_BWD_CASE(8) MF_BWD_CASE(7) MF_BWD_CASE(6) MF_BWD_CASE(5)
            MF_BWD_CASE(4) MF_BWD_CASE(3) MF_BWD_CASE(2) MF_BWD_CASE(1) MF_BWD_CASE(0)
            if (cnt) MF_BWD_FLUSH();
            MF_STAMP_FINE(1);
        }
        __syncthreads();
        if (A.alpha_out)
            for (int i = tid; i < n; i += MF_THREADS)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) A.alpha_out[(size_t)c * A.n_total + o + i] = av[c * NPAD + i];
        MF_STAMP(8);

        // ---- predictive mean ----
        if (A.xs0 == nullptr && A.grid_sz <= 32) {
            // separable grid: f[py][px] = sum_i Ey[py][i] * (sf alpha_i Ex[px][i]); wave w takes i in [32 w, 32 w + 32)
            const int lr = lane & 15, lg = lane >> 4;
            const int sz = A.grid_sz;
            const double res = A.grid_res;
            double* red = panBase;   // 8 waves x 4 tiles x 256 doubles = 64 KB: aliases the (dead) panel buffers
            double ea[2][8], eb[2][8];
            // grid factors of the 32 training points [ib, ib + 32): Ey[py][i], Ex[px][i]
            auto grid_factors = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pq = 16 * h + lr;
                    const double gq = res * (((double)pq + 0.5) / (double)sz - 0.5);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ib + 4 * s + lg;
                        const bool on = (pq < sz) && (i < n);
                        const double dy = gq - px1[on ? i : 0], dx = gq - px0[on ? i : 0];
                        if (small_grid) {
                            ea[h][s] = on ? gpc_exp_small(cexp * (dy * dy)) : 0.0;    // Ey[py = pq][i]
                            eb[h][s] = on ? gpc_exp_small(cexp * (dx * dx)) : 0.0;    // Ex[px = pq][i]
                        } else {
                            ea[h][s] = on ? gpc_exp_neg(cexp * (dy * dy), T) : 0.0;
                            eb[h][s] = on ? gpc_exp_neg(cexp * (dx * dx), T) : 0.0;
                        }
                    }
                }
            };
            const int ibase = 32 * wave;
            const bool wave_live = ibase < n;
            if (wave_live) grid_factors(ibase);
            for (int c = 0; c < ny; ++c) {
                d4 P[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) P[mt][nl] = d4{0.0, 0.0, 0.0, 0.0};
                // one 32-point chunk per wave covers 256 points; NT = 17 (ny == 1) takes a second trip for the points 256 .. 271
                for (int ib = ibase; ib < n; ib += 32 * MF_WAVES) {
                    if (NT > 16 && ib != ibase) grid_factors(ib);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = ib + 4 * s + lg;
                        // rows between 16 nt and the end of this 32-row group are never written by the solve: whatever the
                        // previous kernel left in LDS there (possibly NaN) must not reach the sum -- select, do not multiply
                        const double al = (i < n) ? sf * av[c * NPAD + i] : 0.0;
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
                    const double kk = gpc_rbf_neg(sf, cexp, px0[i], px1[i], q0, q1, T);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ny) s_[c] += kk * av[c * NPAD + i];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) fs[(size_t)c * m + p] = s_[c];
            }
        }
        MF_STAMP(9);
        MF_STAMP_FLUSH();
        if (timed_out && lane == 0) flag[0] = 2;
        __syncthreads();
        if (tid == 0 && A.status) A.status[patch] = flag[0] ? GPC_STATUS_NAN : GPC_STATUS_OK;
    } while (0);
}

bool dense_mfma_supported(const DenseArgs& a)
{
    // n <= 256 for depth and colour; n <= 272 for the depth plane alone (NT = 17, see the LDS carve); the variance path exports
    // the factor of the n <= 256 shapes only
    return (a.n_max <= MF_NPAD || (a.n_max <= 17 * MF_TS && a.ny == 1 && a.v_star == nullptr)) && (a.ny == 1 || a.ny == 3);
}

template <int NT, bool EXPORT = false>
static int launch_nt(gpc_ctx* ctx, const MfmaParams& g, int grid, const char* name)
{
    const size_t lds = sizeof(double) * (size_t)mf_l_total(NT);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_mfma_kernel<NT, EXPORT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((dense_mfma_kernel<NT, EXPORT>), dim3(grid), dim3(MF_THREADS), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    ctx->last_dense_kernel = name;
    return GPC_OK;
}

// Predictive variance (gaussian_process::predict_measurements, /root/reference/src/gaussian_process.cpp:35-43): the fit runs as
// usual and additionally writes its factor (operand images of every L_ik and of the L_ii^-1) into the context's workspace,
// 272 KB per patch at NT = 16; dense_variance.hip then evaluates V* = k** - ||L^-1 k*||^2 from there.
int dense_mfma_launch(gpc_ctx* ctx, const DenseArgs& a_in)
{
    DenseArgs a = a_in;
    double* v_star = a.v_star;
    a.v_star = nullptr;
    const int nt_max = a.n_max <= 64 ? 4 : a.n_max <= 128 ? 8 : a.n_max <= 192 ? 12 : 16;
    size_t fbytes = 0;
    if (v_star) {
        if (a.sel) return gpc_fail(ctx, GPC_EINVAL, "variance + size-class dispatch is not supported");
        fbytes = sizeof(double) * (size_t)a.P * (nt_max * (nt_max + 1) / 2) * MF_IMG;
        const int rc = gpc_ws_reserve(ctx, fbytes + sizeof(double) * (size_t)a.n_total * a.ny);
        if (rc != GPC_OK) return rc;
        // the variance kernel evaluates K* anyway: it forms the mean from the same tiles, so the fit predicts nothing
        a.m = 0;
        if (!a.alpha_out) a.alpha_out = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + fbytes);
    }
    MfmaParams g;
    g.a = a;
    g.export_L = v_star ? static_cast<double*>(ctx->ws) : nullptr;
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
            static const char* names[MF_NPH] = {"load+gram", "wait ready | B2 (factor)", "trsm | z+publish (factor)", "wait B2 | y_j (factor)",
                                                "diag tile upd | wait tile (factor)", "y rows | diag factor (factor)", "trailing update",
                                                "post-loop", "backward", "predict", "bwd: waits (fine)", "bwd: alpha | sub-diag product (fine)"};
            fprintf(stderr, "[MF_STAMPS] mean cycles per patch, by wave (s_memtime ticks); wave 7 is the factor wave:\n%-36s", "phase");
            for (int w = 0; w < MF_WAVES; ++w) fprintf(stderr, "   wave%d", w);
            fprintf(stderr, "\n");
            for (int q = 0; q < MF_NPH; ++q) {
                fprintf(stderr, "%-36s", names[q]);
                for (int w = 0; w < MF_WAVES; ++w) {
                    double s_ = 0;
                    for (int b = 0; b < grid; ++b) s_ += (double)h[((size_t)b * MF_WAVES + w) * MF_NPH + q];
                    fprintf(stderr, " %7.0f", s_ / grid);
                }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "%-36s", "total");
            for (int w = 0; w < MF_WAVES; ++w) {
                double s_ = 0;
                for (int b = 0; b < grid; ++b)
                    for (int q = 0; q < 10; ++q) s_ += (double)h[((size_t)b * MF_WAVES + w) * MF_NPH + q];
                fprintf(stderr, " %7.0f", s_ / grid);
            }
            fprintf(stderr, "\n");
        }
    } dump{ctx, g.stamps, grid};
#endif
#ifdef MF_TRACE
    GPC_HIP(ctx, hipMalloc(&g.stamps, sizeof(unsigned long long) * (size_t)grid * MF_WAVES * MF_NTR));
    GPC_HIP(ctx, hipMemsetAsync(g.stamps, 0, sizeof(unsigned long long) * (size_t)grid * MF_WAVES * MF_NTR, ctx->stream));
    struct TraceDump {
        gpc_ctx* ctx; unsigned long long* d; int grid;
        ~TraceDump()
        {
            std::vector<unsigned long long> h((size_t)grid * MF_WAVES * MF_NTR);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            std::vector<double> sum((size_t)MF_WAVES * MF_NTR, 0.0), cnt((size_t)MF_WAVES * MF_NTR, 0.0);
            for (int b = 0; b < grid; ++b) {
                unsigned long long t0 = ~0ull;
                for (int q = 0; q < MF_WAVES * MF_NTR; ++q) {
                    const unsigned long long v = h[(size_t)b * MF_WAVES * MF_NTR + q];
                    if (v && v < t0) t0 = v;
                }
                for (int q = 0; q < MF_WAVES * MF_NTR; ++q) {
                    const unsigned long long v = h[(size_t)b * MF_WAVES * MF_NTR + q];
                    if (v) { sum[q] += (double)(v - t0); cnt[q] += 1.0; }
                }
            }
            auto at = [&](int w, int slot) { const size_t q = (size_t)w * MF_NTR + slot; return cnt[q] > 0 ? sum[q] / cnt[q] : -1.0; };
            fprintf(stderr, "[MF_TRACE] mean s_memtime ticks since the first trace point of the patch\n");
            fprintf(stderr, "step | factor: diag0 diag1 ready  sub   tile  end  | wave0: ready trsm  pan   prio  yrow  bulk | wave3: ready trsm  pan   prio  yrow  bulk | wave6: ...\n");
            fprintf(stderr, "step | end of the step's last phase, waves 0..6 | panel complete seen by wave 0\n");
            for (int k = 0; k < 16; ++k) {
                fprintf(stderr, "%4d |", k);
                for (int w = 0; w < 7; ++w) fprintf(stderr, " %6.0f", at(w, 8 * k + 5));
                fprintf(stderr, " | %6.0f\n", at(0, 8 * (k + 1)));
            }
            for (int k = 0; k < 16; ++k) {
                fprintf(stderr, "%4d |", k);
                for (int p = 0; p < 6; ++p) fprintf(stderr, " %6.0f", at(7, 8 * k + p));
                for (int w : {0, 3, 6}) {
                    fprintf(stderr, " |");
                    for (int p = 0; p < 6; ++p) fprintf(stderr, " %6.0f", at(w, 8 * k + p));
                }
                fprintf(stderr, "\n");
            }
        }
    } tdump{ctx, g.stamps, grid};
#endif
    if (v_star) {
        int rc;
        if (nt_max == 4) rc = launch_nt<4, true>(ctx, g, grid, "dense_mfma_nt4 + dense_variance");
        else if (nt_max == 8) rc = launch_nt<8, true>(ctx, g, grid, "dense_mfma_nt8 + dense_variance");
        else if (nt_max == 12) rc = launch_nt<12, true>(ctx, g, grid, "dense_mfma_nt12 + dense_variance");
        else rc = launch_nt<16, true>(ctx, g, grid, "dense_mfma_nt16 + dense_variance");
        if (rc != GPC_OK) return rc;
        DenseArgs av = a;
        av.m = a_in.m;
        return dense_variance_launch(ctx, av, nt_max, g.export_L, a.alpha_out, v_star);
    }
    if (a.n_max <= 64) return launch_nt<4>(ctx, g, grid, "dense_mfma_nt4");
    if (a.n_max <= 128) return launch_nt<8>(ctx, g, grid, "dense_mfma_nt8");
    if (a.n_max <= 192) return launch_nt<12>(ctx, g, grid, "dense_mfma_nt12");
    if (a.n_max <= 256) return launch_nt<16>(ctx, g, grid, "dense_mfma_nt16");
    return launch_nt<17>(ctx, g, grid, "dense_mfma_nt17");
}
