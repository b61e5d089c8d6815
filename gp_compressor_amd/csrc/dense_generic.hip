// dense_generic.hip -- generic batched dense-GP kernel: any n <= GPC_MAX_POINTS, ny in {1,3}, optional variance.
//
// Replaces, per patch, gaussian_process::add_measurements + predict_measurements
// (/root/reference/src/gaussian_process.cpp:15-45):  K (+noise, twice when ref_double_noise) -> LLT -> alpha ->
// f* = K*^T alpha [-> V* = k** - |L^-1 k*|^2].
//
// One 256-thread workgroup per patch slot (grid-stride over patches).  The Gram matrix / Cholesky factor of
// the patch lives in a per-workgroup slot of a global-memory workspace that is re-used for every patch the
// workgroup processes, so it stays L2 / Infinity-Cache resident; HBM sees only the 24 n + 8 m bytes of inputs
// and outputs.  Factorisation is a blocked LEFT-looking Cholesky with 16-column panels: each thread owns one
// matrix row of the panel (16 accumulators), the 16 x k block of already-factored rows is staged through LDS
// in chunks and read as LDS broadcasts, the 16 x 16 diagonal block is factored inside wave 0 with cross-lane
// shuffles, and the right-hand sides ride along as `ny` extra matrix rows so that the forward solve L z = y
// falls out of the panel TRSM.  This kernel is the correctness anchor and the path for n > 256; the n <= 256
// bench path is dense_mfma.hip.
#include "gpc_device.h"
#include "gpc_internal.h"

#define GEN_THREADS 256
#define GEN_NB 16
#define GEN_KC 64

struct GenParams {
    DenseArgs a;
    double c_exp;      // (double)(-0.5f) / l_sq
    double pivot_tol;  // GPC_PIVOT_RTOL * (sigmaf_sq + noise)
    double* ws;        // workspace base
    size_t slot;       // doubles per workgroup slot
    int ld;            // leading dimension of the K/L slot (>= n_max + ny)
    int mpad;          // m rounded up to a multiple of 64 (variance scratch row stride)
};

// In-register Cholesky of a w x w diagonal block held one row per lane (lanes 0..w-1 of wave 0, a[jj] = row
// entries).  All 64 lanes of the wave execute it (shuffles).  Returns false in every lane if a pivot is <= 0.
__device__ static inline bool diag_chol_wave(double (&a)[GEN_NB], int w, int lane, double pivot_tol)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < GEN_NB; ++j) {
        if (j < w) {
            double dj = __shfl(a[j], j, 64);
            if (!(dj > pivot_tol)) ok = false;
            double piv = sqrt(dj);
            double lij = a[j] / piv;                 // lane i: L[i][j] (valid for i >= j)
            if (lane == j) lij = piv;
            // lanes >= w of this wave hold OTHER panel rows (or the rhs rows): their accumulators must not change
            if (lane < w) a[j] = lij;
#pragma unroll
            for (int k = j + 1; k < GEN_NB; ++k) {
                double lkj = __shfl(lij, k, 64);     // L[k][j]
                if (k < w && lane < w) a[k] -= lij * lkj;   // only entries with lane >= k are used later
            }
        }
    }
    return ok;
}

__global__ __launch_bounds__(GEN_THREADS) void dense_generic_kernel(GenParams g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DenseArgs& A = g.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ny = A.ny, n_max = A.n_max, ld = g.ld;
    double* T = reinterpret_cast<double*>(smem);             // exp table              64
    double* D = T + 64;                                       // diagonal block 16x17   272
    double* Lrow = D + 272;                                   // staged L rows 16 x KC  1024
    double* tmp = Lrow + GEN_NB * GEN_KC;                     // back-solve partials 16 x 4
    int* flag = reinterpret_cast<int*>(tmp + 64);             // not-SPD flag (2 doubles of room)
    double* xs0l = tmp + 66;                                  // x0, x1: n_max each
    double* xs1l = xs0l + n_max;
    double* al = xs1l + n_max;                                // alpha / rhs: ny * n_max
    double* W = g.ws + (size_t)blockIdx.x * g.slot;
    double* VS = W + (size_t)ld * (size_t)ld;                 // variance scratch: n_max x mpad (only if want_variance)

    gpc_exp_table_init(T);
    const double sf = A.prm.sigmaf_sq, cexp = g.c_exp, noise = A.prm.noise;
    const int m = A.m;

    // (size-class dispatch: the overflow launch behind a class launch sized from a host-side hint works on sel[sel_base .. *sel_count))
    const int n_items = A.sel ? A.sel_count[0] - A.sel_base : A.P;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int patch = A.sel ? A.sel[A.sel_base + item] : item;
        const int o = A.off[patch];
        const int n = A.off[patch + 1] - o;
        double* fs = A.f_star + (size_t)patch * ny * m;
        __syncthreads();  // previous patch fully done with LDS
        if (n <= 0 || n > n_max) {
            // b == 0 analogue: nothing to condition on -> prior mean 0, prior variance k** (gaussian_process.cpp:28-45 with empty X)
            for (int p = tid; p < m * ny; p += GEN_THREADS) fs[p] = (n == 0) ? 0.0 : __builtin_nan("");
            if (A.v_star)
                for (int p = tid; p < m; p += GEN_THREADS) A.v_star[(size_t)patch * m + p] = (n == 0) ? sf : __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = (n == 0) ? GPC_STATUS_OK : GPC_STATUS_NAN;
            continue;
        }
        for (int i = tid; i < n; i += GEN_THREADS) {
            xs0l[i] = A.x0[o + i];
            xs1l[i] = A.x1[o + i];
            for (int c = 0; c < ny; ++c) al[c * n_max + i] = A.y[(size_t)c * A.n_total + o + i];
        }
        if (tid == 0) *flag = 0;
        __syncthreads();

        // ---- Gram matrix, lower triangle + full diagonal blocks; rows n..n+ny-1 hold the right-hand sides ----
        for (int j = 0; j < n; ++j) {
            const int i0 = j & ~(GEN_NB - 1);
            const double xj0 = xs0l[j], xj1 = xs1l[j];
            for (int i = i0 + tid; i < n + ny; i += GEN_THREADS) {
                double v;
                if (i < n) {
                    v = gpc_rbf(sf, cexp, xs0l[i], xs1l[i], xj0, xj1, T);
                    if (i == j) {
                        v += noise;                                  // covariance_matrix(..., training) :59-61
                        if (A.prm.ref_double_noise) v += noise;      // C.diagonal() += sigman_sq       :21
                    }
                } else {
                    v = al[(i - n) * n_max + j];
                }
                W[i + (size_t)j * ld] = v;
            }
        }
        __syncthreads();

        // ---- blocked left-looking Cholesky (+ forward solve on the extra rows) ----
        const int nrows = n + ny;
        bool bad = false;
        for (int c0 = 0; c0 < n; c0 += GEN_NB) {
            const int w = min(GEN_NB, n - c0);
            for (int rbase = c0; rbase < nrows; rbase += GEN_THREADS) {
                const int r = rbase + tid;
                const bool active = r < nrows;
                double acc[GEN_NB];
#pragma unroll
                for (int jj = 0; jj < GEN_NB; ++jj) acc[jj] = (active && jj < w) ? W[r + (size_t)(c0 + jj) * ld] : 0.0;
                for (int k0 = 0; k0 < c0; k0 += GEN_KC) {
                    const int kc = min(GEN_KC, c0 - k0);
                    __syncthreads();
                    for (int e = tid; e < GEN_NB * kc; e += GEN_THREADS) {
                        const int jj = e & (GEN_NB - 1), kk = e >> 4;
                        Lrow[jj + GEN_NB * kk] = (jj < w) ? W[(c0 + jj) + (size_t)(k0 + kk) * ld] : 0.0;
                    }
                    __syncthreads();
                    if (active) {
                        for (int kk = 0; kk < kc; ++kk) {
                            const double lrk = W[r + (size_t)(k0 + kk) * ld];
#pragma unroll
                            for (int jj = 0; jj < GEN_NB; ++jj) acc[jj] -= lrk * Lrow[jj + GEN_NB * kk];
                        }
                    }
                }
                if (rbase == c0) {
                    // rows c0..c0+w-1 are lanes 0..w-1 of wave 0: factor the diagonal block in registers
                    if (wave == 0) {
                        bool ok = diag_chol_wave(acc, w, lane, g.pivot_tol);
                        if (!ok && lane == 0) *flag = 1;
                        if (lane < w) {
#pragma unroll
                            for (int jj = 0; jj < GEN_NB; ++jj) D[lane * 17 + jj] = acc[jj];   // D[i][j] = L[c0+i][c0+j], j <= i
                        }
                    }
                    __syncthreads();
                    bad = (*flag != 0);
                    if (bad) break;
                }
                if (active) {
                    if (r < c0 + w) {
#pragma unroll
                        for (int jj = 0; jj < GEN_NB; ++jj)
                            if (jj <= r - c0) W[r + (size_t)(c0 + jj) * ld] = acc[jj];
                    } else {
                        // x L_dd^T = acc  (row-wise forward substitution against the diagonal block)
#pragma unroll
                        for (int jj = 0; jj < GEN_NB; ++jj) {
                            if (jj < w) {
                                double s = acc[jj];
#pragma unroll
                                for (int c = 0; c < jj; ++c) s -= acc[c] * D[jj * 17 + c];
                                s /= D[jj * 17 + jj];
                                acc[jj] = s;
                                W[r + (size_t)(c0 + jj) * ld] = s;
                            }
                        }
                    }
                }
            }
            if (bad) break;
            __syncthreads();
        }
        if (bad) {
            // Eigen::LLT would flag NumericalIssue; the oracle returns NaN alpha
            for (int p = tid; p < m * ny; p += GEN_THREADS) fs[p] = __builtin_nan("");
            if (A.v_star)
                for (int p = tid; p < m; p += GEN_THREADS) A.v_star[(size_t)patch * m + p] = __builtin_nan("");
            if (A.alpha_out)
                for (int i = tid; i < n * ny; i += GEN_THREADS)
                    A.alpha_out[(size_t)(i / n) * A.n_total + o + (i % n)] = __builtin_nan("");
            if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_NOT_SPD;
            continue;
        }

        // z (forward-solved rhs) sits in rows n..n+ny-1 of W:  z_c[j] = W[n + c + j*ld]
        for (int i = tid; i < n; i += GEN_THREADS)
            for (int c = 0; c < ny; ++c) al[c * n_max + i] = W[(n + c) + (size_t)i * ld];
        __syncthreads();

        // ---- backward solve L^T alpha = z, 16-column blocks from the last to the first ----
        const int nblk = (n + GEN_NB - 1) / GEN_NB;
        for (int b = nblk - 1; b >= 0; --b) {
            const int c0 = b * GEN_NB;
            const int w = min(GEN_NB, n - c0);
            // partial[jj][c] = sum_{r >= c0+16} L[r][c0+jj] * alpha_c[r];  thread = (s = tid%16, jj = tid/16)
            {
                const int s = tid & 15, jj = tid >> 4;
                double part[3] = {0.0, 0.0, 0.0};
                if (jj < w) {
                    for (int r = c0 + GEN_NB + s; r < n; r += 16) {
                        const double l = W[r + (size_t)(c0 + jj) * ld];
                        for (int c = 0; c < ny; ++c) part[c] += l * al[c * n_max + r];
                    }
                }
                for (int c = 0; c < ny; ++c) {
                    double v = part[c];
                    v += __shfl_xor(v, 1, 64);
                    v += __shfl_xor(v, 2, 64);
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 8, 64);
                    if (s == 0) tmp[jj * 4 + c] = v;
                }
                // diagonal block of L into LDS
                const int i = tid & 15, j2 = tid >> 4;
                D[i * 17 + j2] = (i < w && j2 <= i) ? W[(c0 + i) + (size_t)(c0 + j2) * ld] : 0.0;
            }
            __syncthreads();
            if (wave == 0) {
                for (int c = 0; c < ny; ++c) {
                    double t = (lane < w) ? al[c * n_max + c0 + lane] - tmp[lane * 4 + c] : 0.0;
                    for (int i = w - 1; i >= 0; --i) {
                        double ai = __shfl(t, i, 64) / D[i * 17 + i];
                        if (lane == i) t = ai;
                        if (lane < i) t -= D[i * 17 + lane] * ai;
                    }
                    if (lane < w) al[c * n_max + c0 + lane] = t;
                }
            }
            __syncthreads();
        }
        if (A.alpha_out)
            for (int i = tid; i < n; i += GEN_THREADS)
                for (int c = 0; c < ny; ++c) A.alpha_out[(size_t)c * A.n_total + o + i] = al[c * n_max + i];

        // ---- predictive mean f* = K*^T alpha on X* (point-wise list or the reference's sz x sz grid) ----
        for (int p = tid; p < m; p += GEN_THREADS) {
            double q0, q1;
            if (A.xs0) {
                q0 = A.xs0[p];
                q1 = A.xs1[p];
            } else {
                const int gx = p % A.grid_sz, gy = p / A.grid_sz;       // p = y*sz + x (gp_compressor.cpp:320-329)
                q0 = A.grid_res * (((double)gx + 0.5) / (double)A.grid_sz - 0.5);
                q1 = A.grid_res * (((double)gy + 0.5) / (double)A.grid_sz - 0.5);
            }
            double s[3] = {0.0, 0.0, 0.0};
            for (int i = 0; i < n; ++i) {
                const double k = gpc_rbf(sf, cexp, xs0l[i], xs1l[i], q0, q1, T);
                for (int c = 0; c < ny; ++c) s[c] += k * al[c * n_max + i];
            }
            for (int c = 0; c < ny; ++c) fs[(size_t)c * m + p] = s[c];
        }

        // ---- optional predictive variance V*_p = k** - |L^-1 k*_p|^2 (gaussian_process.cpp:35-43) ----
        if (A.v_star) {
            const int mpad = g.mpad;
            for (int pbase = 0; pbase < m; pbase += GEN_THREADS) {
                const int p = pbase + tid;
                const bool active = p < m;
                double q0 = 0.0, q1 = 0.0;
                if (active) {
                    if (A.xs0) {
                        q0 = A.xs0[p];
                        q1 = A.xs1[p];
                    } else {
                        const int gx = p % A.grid_sz, gy = p / A.grid_sz;
                        q0 = A.grid_res * (((double)gx + 0.5) / (double)A.grid_sz - 0.5);
                        q1 = A.grid_res * (((double)gy + 0.5) / (double)A.grid_sz - 0.5);
                    }
                }
                double vv = 0.0;
                for (int c0 = 0; c0 < n; c0 += GEN_NB) {
                    const int w = min(GEN_NB, n - c0);
                    double acc[GEN_NB];
#pragma unroll
                    for (int jj = 0; jj < GEN_NB; ++jj)
                        acc[jj] = (active && jj < w) ? gpc_rbf(sf, cexp, xs0l[c0 + jj], xs1l[c0 + jj], q0, q1, T) : 0.0;
                    for (int k0 = 0; k0 < c0; k0 += GEN_KC) {
                        const int kc = min(GEN_KC, c0 - k0);
                        __syncthreads();
                        for (int e = tid; e < GEN_NB * kc; e += GEN_THREADS) {
                            const int jj = e & (GEN_NB - 1), kk = e >> 4;
                            Lrow[jj + GEN_NB * kk] = (jj < w) ? W[(c0 + jj) + (size_t)(k0 + kk) * ld] : 0.0;
                        }
                        __syncthreads();
                        if (active) {
                            for (int kk = 0; kk < kc; ++kk) {
                                const double vk = VS[(size_t)(k0 + kk) * mpad + p];
#pragma unroll
                                for (int jj = 0; jj < GEN_NB; ++jj) acc[jj] -= vk * Lrow[jj + GEN_NB * kk];
                            }
                        }
                    }
                    __syncthreads();
                    {
                        const int i = tid & 15, j2 = tid >> 4;
                        D[i * 17 + j2] = (i < w && j2 <= i) ? W[(c0 + i) + (size_t)(c0 + j2) * ld] : 0.0;
                    }
                    __syncthreads();
                    if (active) {
#pragma unroll
                        for (int jj = 0; jj < GEN_NB; ++jj) {
                            if (jj < w) {
                                double s = acc[jj];
#pragma unroll
                                for (int c = 0; c < jj; ++c) s -= acc[c] * D[jj * 17 + c];
                                s /= D[jj * 17 + jj];
                                acc[jj] = s;
                                VS[(size_t)(c0 + jj) * mpad + p] = s;
                                vv += s * s;
                            }
                        }
                    }
                }
                if (active) A.v_star[(size_t)patch * m + p] = sf - vv;   // k** = sigmaf_sq * exp(0)
            }
        }
        if (tid == 0 && A.status) A.status[patch] = GPC_STATUS_OK;
    }
}

static size_t gen_lds_bytes(const DenseArgs& a)
{
    return sizeof(double) * (size_t)(64 + 272 + GEN_NB * GEN_KC + 66 + (2 + a.ny) * a.n_max);
}

size_t dense_generic_ws_bytes(const gpc_ctx* ctx, const DenseArgs& a, int* grid_out)
{
    // one slot per resident workgroup; cap the grid so that the workspace stays within a few GB
    const int ld = a.n_max + a.ny;
    const int mpad = (a.m + 63) & ~63;
    size_t slot = (size_t)ld * ld + (a.v_star ? (size_t)a.n_max * mpad : 0);
    const size_t lds = gen_lds_bytes(a);
    int per_cu = (int)((160u * 1024u) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = ctx->num_cus * per_cu;
    const size_t budget = (size_t)8 << 30;
    while (grid > 64 && (size_t)grid * slot * sizeof(double) > budget) grid /= 2;
    if (grid > a.P) grid = a.P;
    if (grid < 1) grid = 1;
    *grid_out = grid;
    return (size_t)grid * slot * sizeof(double);
}

int dense_generic_launch(gpc_ctx* ctx, const DenseArgs& a, int grid, double* ws_override)
{
    GenParams g;
    g.a = a;
    g.c_exp = (double)(-0.5f) / a.prm.l_sq;
    g.pivot_tol = GPC_PIVOT_RTOL * (a.prm.sigmaf_sq + a.prm.noise);
    g.ws = ws_override ? ws_override : static_cast<double*>(ctx->ws);
    g.ld = a.n_max + a.ny;
    g.mpad = (a.m + 63) & ~63;
    g.slot = (size_t)g.ld * g.ld + (a.v_star ? (size_t)a.n_max * g.mpad : 0);
    const size_t lds = gen_lds_bytes(a);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_generic_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(dense_generic_kernel, dim3(grid), dim3(GEN_THREADS), lds, ctx->stream, g);
    GPC_HIP(ctx, hipGetLastError());
    ctx->last_dense_kernel = "dense_generic";
    return GPC_OK;
}
