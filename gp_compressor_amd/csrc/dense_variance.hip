// dense_variance.hip -- predictive variance of the dense GP on the MFMA pipe, n <= 256.
//
// gaussian_process::predict_measurements (/root/reference/src/gaussian_process.cpp:35-43):
//     v = chol.matrixL().solve(K_star);   V_star(j) = squared_exp_distance(x*_j, x*_j) - v.col(j).squaredNorm()
// -- an n x m triangular solve per patch (n^2 m flops: four times the whole fit at n = 256, m = 400), which the reference
// always pays although gp_compressor never reads the result (src/gp_compressor.cpp:333-334).  Round 1 sent every variance
// request to the generic kernel (3 TFLOP/s).  Here the register-tile kernel fits as usual and additionally exports its factor
// as MFMA operand images (dense_mfma.hip, EXPORT instantiation: L_ik for k < i, L_ii^-1 on the diagonal, row-major over the
// lower triangle, 2 KB each), and this kernel solves from there:
//
//   * one 512-thread workgroup per patch; each WAVE owns one block of 16 prediction points at a time (m = 400: 25 blocks, four
//     rounds of eight) and runs the forward substitution of that block entirely in registers: V_i = L_ii^-1 (B_i - sum_{k<i}
//     L_ik V_k), where B_i is evaluated on the fly (K* never exists in memory) and every V_k stays in the C/D register layout
//     of the MFMA that produced it -- which is exactly the B-operand layout of the MFMA that consumes it.  4 MFMAs per tile
//     product, accumulator-chained; the dependency chain of a block is hidden by the SIMD's second wave and by the eight
//     blocks in flight.
//   * the factor is the A operand of every product, the same for all eight waves: it streams through LDS in chunks of eight
//     images (16 KB, double-buffered), loaded once per workgroup and round -- each thread fetches 32 bytes of the next chunk
//     while the waves consume the current one; one barrier per chunk.
//   * squared column norms accumulate in registers; two shuffles combine the four row groups of a lane column at the end.
//   * the mean comes out of the same B_i tiles (f* = K*^T alpha, :32: four FMAs per tile and channel), so the fit kernel
//     skips its own prediction phase when the variance is requested.
// Measured on C2 (8192 x 256, m = 400): fit + mean + variance 8.6 ms = 955 k patches/s, 31.8 TFLOP/s = 0.40 of the FP64 peak
// counting F(n, m) + n^2 m + 2 n m flops per patch (the generic kernel: 91 ms).
#include <cstdlib>

#include "gpc_device.h"
#include "gpc_internal.h"
#include "mfma_tile.h"

#define DV_THREADS 512
#define DV_WAVES 8
#define DV_CH 8   // operand images per LDS chunk

struct VarParams {
    DenseArgs a;
    double c_exp;
    const double* factor;   // [P][NT (NT + 1) / 2][256]
    const double* alpha;    // [ny][n_total]: the fit's weights (the mean f* = K*^T alpha comes out of the same B_i tiles for free)
    double* v_star;         // [P][m]
};

// WV waves per workgroup: 8 (one workgroup per CU) or 4 (two: 25 blocks of 16 points are four rounds of eight with one wave busy in
// the last -- 78 % of the MFMA slots -- but seven rounds of four with 89 %, and the two workgroups of a CU cover each other's barriers)
template <int NT, int WV = 8>
__global__ __launch_bounds__(64 * WV, 2) void dense_variance_kernel(VarParams g)
{
    constexpr int DVT = 64 * WV;                                  // threads
    constexpr int NPRE = (DV_CH * MF_IMG) / DVT / 4;              // d4's of a chunk per thread
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int NCHUNK = (NTILES + DV_CH - 1) / DV_CH;
    __shared__ __attribute__((aligned(16))) double T[GPC_EXP_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) double px0[NT * MF_TS], px1[NT * MF_TS], al[3][NT * MF_TS];
    __shared__ __attribute__((aligned(16))) double Lbuf[2][DV_CH * MF_IMG];

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int m = A.m;
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp;
    gpc_exp_table_init(T);

    const int patch = blockIdx.x;
    const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
    const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny);
    double* vs = g.v_star + (size_t)patch * m;
    double* fs = A.f_star + (size_t)patch * ny * m;
    const int st = A.status ? __builtin_amdgcn_readfirstlane(A.status[patch]) : GPC_STATUS_OK;
    if (n <= 0 || n > NT * MF_TS || st != GPC_STATUS_OK) {
        // no training points: f* = 0, v = sigma_f^2 (the prior); a failed fit: NaN
        for (int p = tid; p < m; p += DVT) vs[p] = (n == 0) ? sf : __builtin_nan("");
        for (int p = tid; p < m * ny; p += DVT) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
        return;
    }
    const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);
    // Small-argument regime (dense_mfma.hip): when |c| d^2 <= 2^-5 for every (training point, prediction point) pair of the patch the
    // K* tiles take the degree-7 polynomial instead of the table-driven exponential -- half the VALU work of a block's 64 evaluations
    // per lane, on the pipe the MFMAs need.  Bound: max-norm distances of the points and of X* from the patch's first point.
    __shared__ unsigned long long ext_bits[2];
    if (tid < 2) ext_bits[tid] = 0ull;
    __syncthreads();
    const double xo0 = A.x0[o], xo1 = A.x1[o];
    double dev_p = 0.0, dev_q = 0.0;
    for (int i = tid; i < NT * MF_TS; i += DVT) {
        const bool live = i < n;
        const double q0 = live ? A.x0[o + i] : xo0, q1 = live ? A.x1[o + i] : xo1;
        dev_p = __builtin_fmax(dev_p, __builtin_fmax(__builtin_fabs(q0 - xo0), __builtin_fabs(q1 - xo1)));
        px0[i] = live ? q0 : 0.0;
        px1[i] = live ? q1 : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) al[c][i] = (c < ny && i < n) ? g.alpha[(size_t)c * A.n_total + o + i] : 0.0;
    }
    for (int p = tid; p < m; p += DVT)
        dev_q = __builtin_fmax(dev_q, __builtin_fmax(__builtin_fabs(A.xs0[p] - xo0), __builtin_fabs(A.xs1[p] - xo1)));
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) {
        dev_p = __builtin_fmax(dev_p, __shfl_xor(dev_p, o_, 64));
        dev_q = __builtin_fmax(dev_q, __shfl_xor(dev_q, o_, 64));
    }
    if (lane == 0) {     // non-negative doubles order like their bits; a NaN extent fails the test below
        atomicMax(&ext_bits[0], (unsigned long long)__double_as_longlong(dev_p));
        atomicMax(&ext_bits[1], (unsigned long long)__double_as_longlong(dev_q));
    }
    __syncthreads();
    bool small_k;
    {
        const double r = __longlong_as_double((long long)ext_bits[0]) + __longlong_as_double((long long)ext_bits[1]);
        small_k = __builtin_amdgcn_readfirstlane((int)(-cexp * (2.0 * r * r) <= GPC_EXP_SMALL_MAX)) != 0 && !(dev_p != dev_p) && !(dev_q != dev_q);
    }
    const double* F = g.factor + (size_t)patch * NTILES * MF_IMG;
    // this thread's 32 bytes of a chunk: chunk c = images [c DV_CH, (c + 1) DV_CH) = 2048 doubles, 4 per thread
    const int my4 = tid * 4;                                   // + u * 4 DVT, u < NPRE
    const int nblk = (m + MF_TS - 1) / MF_TS;
    const int rounds = (nblk + WV - 1) / WV;
    // last chunk that rows < nt touch (the stream is consumed in order; everything behind it is never read)
    const int last_pos = (nt * (nt + 1)) / 2 - 1;
    const int last_chunk = __builtin_amdgcn_readfirstlane(last_pos / DV_CH);

    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = rd * WV + wave;
        const bool active = blk < nblk;
        const int q = MF_TS * blk + lr;                 // this lane's prediction point (column of the block)
        const bool qv = active && q < m;
        const double gx0 = qv ? A.xs0[q] : 0.0, gx1 = qv ? A.xs1[q] : 0.0;
        d4 V[NT];
        double nrm = 0.0, fm[3] = {0.0, 0.0, 0.0};
        d4 pre[NPRE];
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = d4{0.0, 0.0, 0.0, 0.0};
        __syncthreads();                                // previous round (or the px load) is complete; Lbuf may be rewritten
        // chunk 0 -> buffer 0, chunk 1 in flight
        {
#pragma unroll
            for (int u = 0; u < NPRE; ++u) *reinterpret_cast<d4*>(&Lbuf[0][my4 + u * 4 * DVT]) = *reinterpret_cast<const d4*>(F + my4 + u * 4 * DVT);
            const int c1 = min(1, NCHUNK - 1);
#pragma unroll
            for (int u = 0; u < NPRE; ++u) pre[u] = *reinterpret_cast<const d4*>(F + (size_t)c1 * DV_CH * MF_IMG + my4 + u * 4 * DVT);
        }
        __syncthreads();
        if (!active) {
            // no block left for this wave in the last round: keep feeding the LDS stream (same barriers, same order) but stay
            // off the MFMA pipe, which the SIMD's other wave may be using
            for (int c = 1; c <= last_chunk; ++c) {
#pragma unroll
                for (int u = 0; u < NPRE; ++u) *reinterpret_cast<d4*>(&Lbuf[c & 1][my4 + u * 4 * DVT]) = pre[u];
                const int cn = min(c + 1, last_chunk);
#pragma unroll
                for (int u = 0; u < NPRE; ++u) pre[u] = *reinterpret_cast<const d4*>(F + (size_t)cn * DV_CH * MF_IMG + my4 + u * 4 * DVT);
                __syncthreads();
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (i < nt) {
                // B_i: K*(rows 16 i + lg + 4 r, column q), straight into the accumulator; rows beyond n are padding -> 0
                d4 acc;
                // (evaluated for every lane -- padded points and columns are finite numbers -- and SELECTED: a mask around the
                // exponential becomes a branch per element)
                if (small_k) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = MF_TS * i + lg + 4 * r;
                        acc[r] = gpc_rbf_small(sf, cexp, px0[p], px1[p], gx0, gx1);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = MF_TS * i + lg + 4 * r;
                        acc[r] = gpc_rbf_neg(sf, cexp, px0[p], px1[p], gx0, gx1, T);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = MF_TS * i + lg + 4 * r;
                    acc[r] = (qv && p < n) ? acc[r] : 0.0;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ny) fm[c] = __builtin_fma(acc[r], al[c][p], fm[c]);      // f* = K*^T alpha (:32), same tile
                }
#pragma unroll
                for (int k = 0; k <= i; ++k) {
                    const int pos = (i * (i + 1)) / 2 + k;          // compile-time after unrolling
                    const int c = pos / DV_CH, slot = pos % DV_CH;
                    if (slot == 0 && c > 0) {
                        // entering chunk c: publish it (prefetched during chunk c-1) and start fetching chunk c+1
#pragma unroll
                        for (int u = 0; u < NPRE; ++u) *reinterpret_cast<d4*>(&Lbuf[c & 1][my4 + u * 4 * DVT]) = pre[u];
                        const int cn = min(c + 1, last_chunk);
#pragma unroll
                        for (int u = 0; u < NPRE; ++u) pre[u] = *reinterpret_cast<const d4*>(F + (size_t)cn * DV_CH * MF_IMG + my4 + u * 4 * DVT);
                        __syncthreads();
                    }
                    const d4 img = mf_img_load(&Lbuf[c & 1][slot * MF_IMG], lane);
                    if (k < i) {
                        // acc -= L_ik V_k   (blgp = 1: NEG(A))
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(img[s], V[k][s], acc, 0, 0, 1);
                    } else {
                        // V_i = L_ii^-1 acc: four independent products, tree sum
                        // V_i = L_ii^-1 acc: accumulator-chained (no VALU add tree behind the MFMA -> VALU hazard, 24 registers less)
                        d4 D = __builtin_amdgcn_mfma_f64_16x16x4f64(img[0], acc[0], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f64_16x16x4f64(img[1], acc[1], D, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f64_16x16x4f64(img[2], acc[2], D, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f64_16x16x4f64(img[3], acc[3], D, 0, 0, 0);
                        V[i] = D;
                        nrm += (V[i][0] * V[i][0] + V[i][1] * V[i][1]) + (V[i][2] * V[i][2] + V[i][3] * V[i][3]);
                    }
                }
            }
        }
        // the four row groups (lg) of a column
        nrm += __shfl_xor(nrm, 16, 64);
        nrm += __shfl_xor(nrm, 32, 64);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fm[c] += __shfl_xor(fm[c], 16, 64);
            fm[c] += __shfl_xor(fm[c], 32, 64);
        }
        if (qv && lg == 0) {
            vs[q] = sf - nrm;                           // squared_exp_distance(x*, x*) = sigma_f^2 (no noise term, :40)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < ny) fs[(size_t)c * m + q] = fm[c];
        }
    }
}

template <int NT>
static int var_launch_t(gpc_ctx* ctx, const VarParams& g, int grid)
{
    // four waves per workgroup (two workgroups per CU) when that fills the rounds markedly better: m = 400 is 25 blocks -- 0.78 of
    // the wave slots in rounds of eight, 0.89 in rounds of four (C2 + variance 8.22 -> 8.06 ms; the factor is streamed 7 times
    // instead of 4, which eats most of the gain)
    const int nblk = (g.a.m + MF_TS - 1) / MF_TS;
    const double eff8 = (double)nblk / (8 * ((nblk + 7) / 8)), eff4 = (double)nblk / (4 * ((nblk + 3) / 4));
    const bool w4 = getenv("GPC_VAR_W4") ? atoi(getenv("GPC_VAR_W4")) != 0 : eff4 > eff8 + 0.08;
    if (w4) hipLaunchKernelGGL((dense_variance_kernel<NT, 4>), dim3(grid), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((dense_variance_kernel<NT, 8>), dim3(grid), dim3(DV_THREADS), 0, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

int dense_variance_launch(gpc_ctx* ctx, const DenseArgs& a, int nt_max, const double* factor, const double* alpha, double* v_star)
{
    if (a.P == 0 || a.m == 0) return GPC_OK;
    if (!a.xs0 || !a.xs1) return gpc_fail(ctx, GPC_EINVAL, "the predictive variance needs point-wise X*");
    VarParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.factor = factor;
    g.alpha = alpha;
    g.v_star = v_star;
    switch (nt_max) {
        case 4: return var_launch_t<4>(ctx, g, a.P);
        case 8: return var_launch_t<8>(ctx, g, a.P);
        case 12: return var_launch_t<12>(ctx, g, a.P);
        default: return var_launch_t<16>(ctx, g, a.P);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same solve for 256 < n <= 1024, from the tiled kernel's factor (dense_mfma_big.hip keeps the slot of EVERY patch when the
// variance is requested: tiles (i, k) at (i * ntw + k) * 256, the L_kk^-1 images behind them).  A block of 16 prediction points
// now has up to 64 V tiles -- 512 registers -- so the forward substitution runs in SUPER-ROWS of 16 tile rows: the 16
// residual / V tiles of the current super-row live in registers, finished super-rows go to a per-wave scratch in global memory
// (L2-resident: written once, read once per later super-row and column, each tile then serving 16 products from registers),
// and the factor streams through LDS exactly as above, in the order of the traversal:
//     for I:  for J < I:  for kk: V_(16J+kk) <- scratch;  for ii: R_ii -= L_(16I+ii)(16J+kk) V        (column-major in the block)
//             for ii:  for kk < ii: R_ii -= L_(16I+ii)(16I+kk) R_kk;   R_ii <- L_ii^-1 R_ii           (row-major, diagonal block)
// stream position -> tile is a closed form (dv_stream_tile), evaluated by the wave that fetches the tile.
struct VarBigParams {
    DenseArgs a;
    double c_exp;
    const double* ws;       // factor slots, one per patch
    size_t slot;            // doubles per slot
    int ntw;                // tile columns of a slot
    const double* alpha;    // [ny][n_total]
    double* scratch;        // [gridDim.x][8 waves][ntw][256]
    double* v_star;         // [P][m]
};

// super-row I holds rI = min(16, nt - 16 I) tile rows; its part of the stream: I off-diagonal blocks of 16 columns x rI rows
// (column-major), then the diagonal block row-major with L_ii^-1 in the diagonal position
__device__ static __forceinline__ int dv_stream_len(int nt)
{
    int len = 0;
    for (int I = 0; 16 * I < nt; ++I) {
        const int rI = min(16, nt - 16 * I);
        len += 16 * rI * I + (rI * (rI + 1)) / 2;
    }
    return len;
}
// offset (in doubles, relative to the patch's slot) of the image at stream position p
__device__ static __forceinline__ size_t dv_stream_tile(int p, int nt, int ntw)
{
    int I = 0, base = 0;
    for (;; ++I) {
        const int rI = min(16, nt - 16 * I);
        const int len = 16 * rI * I + (rI * (rI + 1)) / 2;
        if (p < base + len || 16 * (I + 1) >= nt) break;
        base += len;
    }
    const int rI = min(16, nt - 16 * I);
    int q = p - base;
    const int offd = 16 * rI * I;
    int i, k;
    if (q < offd) {
        const int J = q / (16 * rI), w = q - J * (16 * rI);
        k = 16 * J + w / rI;
        i = 16 * I + w % rI;
    } else {
        q -= offd;
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= q) ++ii;
        const int kk = q - (ii * (ii + 1)) / 2;
        i = 16 * I + ii;
        k = 16 * I + kk;
        if (kk == ii) return ((size_t)(ntw + 1) * ntw + ntw + i) * MF_IMG;        // L_ii^-1 image
    }
    return ((size_t)i * ntw + k) * MF_IMG;
}

#define DVB_NPAD 1024
__global__ __launch_bounds__(DV_THREADS, 2) void dense_variance_big_kernel(VarBigParams g)
{
    extern __shared__ __attribute__((aligned(16))) char dvb_smem[];
    double* T = reinterpret_cast<double*>(dvb_smem);        // 64
    double* px0 = T + 64;                                   // DVB_NPAD
    double* px1 = px0 + DVB_NPAD;
    double* al = px1 + DVB_NPAD;                            // 3 x DVB_NPAD
    double* Lbuf = al + 3 * DVB_NPAD;                       // 2 x DV_CH images

    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int m = A.m, ntw = g.ntw;
    const int ny = __builtin_amdgcn_readfirstlane(A.ny);
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp;
    gpc_exp_table_init(T);
    double* vs = g.scratch + ((size_t)blockIdx.x * DV_WAVES + wave) * (size_t)ntw * MF_IMG;
    const int my4 = lane * 4;                               // this lane's 32 bytes of the image its wave fetches

    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int o = __builtin_amdgcn_readfirstlane(A.off[patch]);
        const int n = __builtin_amdgcn_readfirstlane(A.off[patch + 1]) - o;
        double* vst = g.v_star + (size_t)patch * m;
        double* fs = A.f_star + (size_t)patch * ny * m;
        const int st = A.status ? __builtin_amdgcn_readfirstlane(A.status[patch]) : GPC_STATUS_OK;
        __syncthreads();                                    // previous patch is done with the LDS vectors
        if (n <= 0 || n > MF_TS * ntw || st != GPC_STATUS_OK) {
            for (int p = tid; p < m; p += DV_THREADS) vst[p] = (n == 0) ? sf : __builtin_nan("");
            for (int p = tid; p < m * ny; p += DV_THREADS) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            continue;
        }
        const int nt = __builtin_amdgcn_readfirstlane((n + MF_TS - 1) / MF_TS);
        for (int i = tid; i < MF_TS * nt; i += DV_THREADS) {
            px0[i] = (i < n) ? A.x0[o + i] : 0.0;
            px1[i] = (i < n) ? A.x1[o + i] : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) al[c * DVB_NPAD + i] = (c < ny && i < n) ? g.alpha[(size_t)c * A.n_total + o + i] : 0.0;
        }
        const double* F = g.ws + (size_t)patch * g.slot;
        const int slen = __builtin_amdgcn_readfirstlane(dv_stream_len(nt));
        const int last_chunk = (slen - 1) / DV_CH;
        const int nblk = (m + MF_TS - 1) / MF_TS;
        const int rounds = (nblk + DV_WAVES - 1) / DV_WAVES;
        const int SR = (nt + 15) / 16;

        for (int rd = 0; rd < rounds; ++rd) {
            const int blk = rd * DV_WAVES + wave;
            const bool active = blk < nblk;
            const int q = MF_TS * blk + lr;
            const bool qv = active && q < m;
            const double gx0 = qv ? A.xs0[q] : 0.0, gx1 = qv ? A.xs1[q] : 0.0;
            double nrm = 0.0, fm[3] = {0.0, 0.0, 0.0};
            d4 pre;
            __syncthreads();                                // previous round (or the vector load) is complete
            // chunk c = stream positions [8 c, 8 c + 8); wave w fetches position 8 c + w (clamped to the end of the stream)
            auto fetch = [&](int c) __attribute__((always_inline)) {
                const int p = min(c * DV_CH + wave, slen - 1);
                const size_t offd = dv_stream_tile(p, nt, ntw);
                return *reinterpret_cast<const d4*>(F + offd + ((size_t)(lane >> 5) * 128 + (lane & 31) * 4));
            };
            // (an image is 2 planes of 128 doubles; lane l of the fetching wave moves doubles [4 l', 4 l' + 4) of plane l >> 5)
            auto publish = [&](int c, d4 v) __attribute__((always_inline)) {
                *reinterpret_cast<d4*>(Lbuf + (size_t)(c & 1) * DV_CH * MF_IMG + wave * MF_IMG + (lane >> 5) * 128 + (lane & 31) * 4) = v;
            };
            publish(0, fetch(0));
            pre = fetch(min(1, last_chunk));
            __syncthreads();
            int p = 0;                                      // stream position of the next image to consume (wave-uniform)
#define DVB_NEXT_IMG(img)                                                                                            \
    do {                                                                                                             \
        if ((p & (DV_CH - 1)) == 0 && p > 0) {                                                                       \
            const int c_ = p / DV_CH;                                                                                \
            publish(c_, pre);                                                                                        \
            pre = fetch(min(c_ + 1, last_chunk));                                                                    \
            __syncthreads();                                                                                         \
        }                                                                                                            \
        img = mf_img_load(Lbuf + (size_t)((p / DV_CH) & 1) * DV_CH * MF_IMG + (p & (DV_CH - 1)) * MF_IMG, lane);     \
        ++p;                                                                                                         \
    } while (0)
            for (int I = 0; I < SR; ++I) {
                const int rI = min(16, nt - 16 * I);
                d4 R[16];
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) {
                    R[ii] = d4{0.0, 0.0, 0.0, 0.0};
                    if (ii < rI) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int pt = MF_TS * (16 * I + ii) + lg + 4 * r;
                            const double kv = (qv && pt < n) ? gpc_rbf_neg(sf, cexp, px0[pt], px1[pt], gx0, gx1, T) : 0.0;
                            R[ii][r] = kv;
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (c < ny) fm[c] = __builtin_fma(kv, al[c * DVB_NPAD + pt], fm[c]);
                        }
                    }
                }
                // older super-rows: V tiles back from the scratch, one at a time, each serving the rI rows of this super-row
                for (int J = 0; J < I; ++J) {
                    d4 Vn = *reinterpret_cast<const d4*>(vs + (size_t)(16 * J) * MF_IMG + my4);
                    for (int kk = 0; kk < 16; ++kk) {
                        const d4 Vk = Vn;
                        const int kn = min(16 * J + kk + 1, 16 * I - 1);
                        Vn = *reinterpret_cast<const d4*>(vs + (size_t)kn * MF_IMG + my4);        // next tile in flight
#pragma unroll
                        for (int ii = 0; ii < 16; ++ii) {
                            if (ii < rI) {
                                d4 img;
                                DVB_NEXT_IMG(img);
                                if (active) {
#pragma unroll
                                    for (int s = 0; s < 4; ++s) R[ii] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[s], Vk[s], R[ii], 0, 0, 1);
                                }
                            }
                        }
                    }
                }
                // the diagonal block
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) {
                    if (ii < rI) {
#pragma unroll
                        for (int kk = 0; kk < ii; ++kk) {
                            d4 img;
                            DVB_NEXT_IMG(img);
                            if (active) {
#pragma unroll
                                for (int s = 0; s < 4; ++s) R[ii] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[s], R[kk][s], R[ii], 0, 0, 1);
                            }
                        }
                        d4 img;
                        DVB_NEXT_IMG(img);
                        if (active) {
                            const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
                            const d4 D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(img[0], R[ii][0], z4, 0, 0, 0);
                            const d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(img[1], R[ii][1], z4, 0, 0, 0);
                            const d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(img[2], R[ii][2], z4, 0, 0, 0);
                            const d4 D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(img[3], R[ii][3], z4, 0, 0, 0);
                            R[ii] = (D0 + D1) + (D2 + D3);
                            nrm += (R[ii][0] * R[ii][0] + R[ii][1] * R[ii][1]) + (R[ii][2] * R[ii][2] + R[ii][3] * R[ii][3]);
                            if (I + 1 < SR) *reinterpret_cast<d4*>(vs + (size_t)(16 * I + ii) * MF_IMG + my4) = R[ii];
                        }
                    }
                }
            }
#undef DVB_NEXT_IMG
            nrm += __shfl_xor(nrm, 16, 64);
            nrm += __shfl_xor(nrm, 32, 64);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                fm[c] += __shfl_xor(fm[c], 16, 64);
                fm[c] += __shfl_xor(fm[c], 32, 64);
            }
            if (qv && lg == 0) {
                vst[q] = sf - nrm;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ny) fs[(size_t)c * m + q] = fm[c];
            }
        }
    }
}

static int dvb_grid(const gpc_ctx* ctx, int P) { return P < ctx->num_cus ? P : ctx->num_cus; }

size_t dense_variance_big_scratch_doubles(const gpc_ctx* ctx, int ntw)
{
    return (size_t)ctx->num_cus * DV_WAVES * (size_t)ntw * MF_IMG;
}

int dense_variance_big_launch(gpc_ctx* ctx, const DenseArgs& a, int ntw, const double* ws, size_t slot, const double* alpha,
                              double* scratch, double* v_star)
{
    if (a.P == 0 || a.m == 0) return GPC_OK;
    if (!a.xs0 || !a.xs1) return gpc_fail(ctx, GPC_EINVAL, "the predictive variance needs point-wise X*");
    VarBigParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.ws = ws; g.slot = slot; g.ntw = ntw; g.alpha = alpha; g.scratch = scratch; g.v_star = v_star;
    const size_t lds = sizeof(double) * (size_t)(64 + 5 * DVB_NPAD + 2 * DV_CH * MF_IMG);
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_variance_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    hipLaunchKernelGGL(dense_variance_big_kernel, dim3(dvb_grid(ctx, a.P)), dim3(DV_THREADS), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}
