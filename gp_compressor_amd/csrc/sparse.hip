// sparse.hip -- batched online sparse GP (Csato-Opper) : sparse_gp<rbf_kernel, gaussian_noise> and
// sparse_gp_field<rbf_kernel, gaussian_noise_3d> of the reference, one patch per workgroup.
//
// Follows /root/reference/src/sparse_gp.hpp:89-249 (add), :252-295 (delete_bv), :299-351 (predict) and the
// field variant /root/reference/src/sparse_gp_field.hpp:59-215, 219-263, 268-320.  The recursion is strictly
// sequential in the points of one patch and data-dependent (sparse vs full update, capacity / geometric
// deletions), so the parallelism is: patches across workgroups, and inside a patch the O(b) kernel vector, the
// two O(b^2) mat-vecs (C k, Q k) and the O(b^2) rank-1 updates across the 256 threads of the workgroup.  All
// branch decisions are taken from values every thread computes identically (LDS-reduced), so control flow is
// workgroup-uniform.
//
// State layout in HBM (persistent across calls, /root/reference/src/gp_mapping.cpp:338-339 keeps adding to
// trained GPs):  alpha [P][ny][ld], C [P][ld][ld], Q [P][ld][ld] column-major like Eigen, BV [P][ld][2] (AoS,
// = Eigen 2 x b column-major), b [P], total_count [P];  ld = capacity + 1 rounded up to 16 (a full update may hold capacity+1
// basis vectors until the deletion that follows it), or GPC_MAX_BV when capacity == -1.
#include <cstdlib>
#include <vector>

#include "gpc_device.h"
#include "gpc_internal.h"

#define SP_THREADS 256

struct gpc_sparse {
    gpc_ctx* ctx;
    gpc_params prm;
    int P, ny, ld;
    double *alpha, *C, *Q, *BV;
    int32_t *b, *count, *stat;
    int32_t* done_it;   // P: hand-over between the phases of an add call (allocated with the object)
    int32_t* list;      // P + 4: work list of the phases after the rows phase, then its length and three ticket counters
    uint8_t* trace;     // diagnostic (gpc_sparse_set_trace): device buffer for the decision bytes of the next add calls, or nullptr
};

struct SpState {
    double *alpha, *C, *Q, *BV;   // this patch
    int ld, ny;                   // ld: stride of the alpha planes (and of the LDS vectors that go with them)
    int ldm;                      // stride of C and Q (== ld in HBM; SP_BMAX when the small-basis kernel keeps them in LDS)
};

// The add kernel and its helpers run with 256 threads per patch, or with 64 (one wave per patch) when capacity <= 64 lets a
// single wave cover every row: SP_NTH is the launch's workgroup size.  Both shapes produce the same bits (the reductions
// and the quarter-wise mat-vec sums are laid out identically); the narrow one has no cross-wave barriers and keeps four
// times as many patches in flight, which is what the small-basis regime needs.
#define SP_NTH ((int)blockDim.x)
#ifndef SP_BMAX
#define SP_BMAX 24   // resident block of the small-basis add kernel (see sparse_add_kernel)
#endif
#define SP_BMID 48   // ... and of its second instance, which takes the patches that have outgrown SP_BMAX (round 4)
// ---- block-wide helpers (all SP_NTH threads call) -------------------------------------------------------

// (every control used here -- quad permutes, row mirrors, row rotate -- has a source lane for every lane, so the destination's previous
// value never shows: the mov form with bound_ctrl spares the `v_mov_b32 dst, 0` that update_dpp(0, ...) puts in front of each of the two
// halves of a double -- 3 instead of 5 VALU operations per step of a row sum, in kernels whose time is their VALU count)
template <int CTRL>
__device__ static __forceinline__ int sp_dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ static __forceinline__ double sp_dpp(double v)
{
    return __hiloint2double(sp_dpp_i<CTRL>(__double2hiint(v)), sp_dpp_i<CTRL>(__double2loint(v)));
}
__device__ static __forceinline__ double sp_readlane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Sum over the 64 lanes of a wave without LDS traffic (the shuffle form costs two ds_bpermute and an LDS round trip per step, six
// steps per sum, eight sums per point: a fifth of the instructions and most of the latency of a point in the small-basis regime).
// Inside a row of 16 the exchanges are symmetric -- quad xor 1, quad xor 2, half-row mirror, row mirror: both partners add the same
// two numbers -- so the 16 lanes of a row end with the same bits; the four row totals then go through SGPRs and are added in one
// fixed order by every lane: the result is uniform by construction (a lane-dependent association would let `gamma < eps_tol`
// diverge inside a wave).
__device__ static __forceinline__ double sp_wave_sum_dpp(double v)
{
    v += sp_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += sp_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += sp_dpp<0x141>(v);     // row_half_mirror
    v += sp_dpp<0x140>(v);     // row_mirror
    return (sp_readlane(v, 0) + sp_readlane(v, 16)) + (sp_readlane(v, 32) + sp_readlane(v, 48));
}

// block-wide sums of the first N (<= 4) entries of v; results broadcast to every thread (the other entries are left alone: a wave
// sum is ~25 instructions, and the callers need 2 + ny of the 8 slots they carry)
template <int N>
__device__ static inline void sp_block_sum4(double (&v)[4], double* scratch /*4*4 doubles*/)
{
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = sp_wave_sum_dpp(v[q]);
    if (SP_NTH == 64) {
        // one-wave shape: what the four-wave layout below computes with three absent waves contributing +0.0 (x + 0.0 is not x for -0.0)
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = ((v[q] + 0.0) + 0.0) + 0.0;
        return;
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) scratch[w * 4 + q] = v[q];
        if (w == 0)                                   // two-wave shape: the absent waves contribute the +0.0 they would have summed
            for (int w2 = SP_NTH >> 6; w2 < 4; ++w2)
#pragma unroll
                for (int q = 0; q < N; ++q) scratch[w2 * 4 + q] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = scratch[q] + scratch[4 + q] + scratch[8 + q] + scratch[12 + q];
}

// argmin with first-index tie break (the reference scans i ascending with a strict '<').  The order is total (value, then index, NaN
// last), so the winner does not depend on the shape of the reduction; rows by DPP, the four row winners through SGPRs.
#define SP_TAKE(ov, oi, val, idx) ((ov) < (val) || ((ov) == (val) && (oi) < (idx)) || ((val) != (val) && (ov) == (ov)))
__device__ static inline void sp_block_argmin(double& val, int& idx, double* sval, int* sidx)
{
#define SP_ARGMIN_STEP(CTRL)                                                                                          \
    do {                                                                                                             \
        const double ov = sp_dpp<CTRL>(val);                                                                         \
        const int oi = sp_dpp_i<CTRL>(idx);                                                                          \
        if (SP_TAKE(ov, oi, val, idx)) { val = ov; idx = oi; }                                                       \
    } while (0)
    SP_ARGMIN_STEP(0xB1);
    SP_ARGMIN_STEP(0x4E);
    SP_ARGMIN_STEP(0x141);
    SP_ARGMIN_STEP(0x140);
#undef SP_ARGMIN_STEP
    {
        double bv = sp_readlane(val, 0);
        int bi = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const double ov = sp_readlane(val, 16 * r);
            const int oi = __builtin_amdgcn_readlane(idx, 16 * r);
            if (SP_TAKE(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        val = bv;
        idx = bi;
    }
    if (SP_NTH == 64) return;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sval[w] = val; sidx[w] = idx; }
    __syncthreads();
    val = sval[0];
    idx = sidx[0];
    for (int q = 1; q < SP_NTH / 64; ++q) {
        double ov = sval[q];
        int oi = sidx[q];
        if (SP_TAKE(ov, oi, val, idx)) { val = ov; idx = oi; }
    }
}

// Read-modify-write pass over the nb x nb blocks of C and Q: each thread takes SP_RMW elements per trip, ALL their loads
// first, then the arithmetic and the stores.  Written element by element, a store to C followed by the next load from C
// may alias as far as the compiler can tell, so every iteration waited out a full memory round trip with one load in
// flight per thread (the update was 50-60 % of the time of a point).  f(i, j, c, q) updates the two values in place; the
// element order does not matter (they are independent).
#define SP_RMW 8
template <int RB = SP_RMW, class F>
__device__ static inline void sp_rmw_cq(double* C, double* Q, int ld, int nb, F f)
{
    const int nn = nb * nb;
    for (int e0 = threadIdx.x; e0 < nn; e0 += SP_NTH * RB) {
        double c[RB], q[RB];
        int ii[RB], jj[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int e = e0 + u * SP_NTH;
            const int ec = e < nn ? e : e0;                      // clamped: the load is unconditional
            ii[u] = ec % nb;
            jj[u] = ec / nb;
            c[u] = C[ii[u] + (size_t)jj[u] * ld];
            q[u] = Q[ii[u] + (size_t)jj[u] * ld];
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            if (e0 + u * SP_NTH < nn) {
                f(ii[u], jj[u], c[u], q[u]);
                C[ii[u] + (size_t)jj[u] * ld] = c[u];
                Q[ii[u] + (size_t)jj[u] * ld] = q[u];
            }
        }
    }
}

// The same pass in the mat-vec's own traversal -- wave w owns the column quarter w, its lanes the rows, columns ascending --
// which lets it hand the NEXT point's mat-vecs C' k' and Q' k' (k' = kvn, evaluated against the basis as it stands after
// this update) out of the values it is about to store: the partial sums land in `pnext` exactly where the stand-alone
// mat-vec of the next iteration would have put them, the same numbers bit for bit, and C and Q are not read a second time
// (a third of the traffic of a point).
#ifndef SP_RMW_NEXT
#define SP_RMW_NEXT 8
#endif
template <bool WRITE_Q = true, int RB = SP_RMW_NEXT, class F>
__device__ static inline void sp_rmw_cq_next(double* C, double* Q, int ld, int lv, int nb, const double* kvn, double* pnext, F f)
{
    const int lane = threadIdx.x & 63;
    if (SP_NTH == 64 && nb <= 32 && (ld == SP_BMAX || ld == SP_BMID)) {     // (ld == SP_BMAX / SP_BMID: C and Q are the LDS blocks of a one-wave kernel)
        // One wave and a small basis (the reference's default hyper-parameters keep it around 13): with a lane per row most of the
        // wave idles and the four column quarters run one after the other -- ~30 instructions per column, 13 columns, every point.
        // Here the lanes form 4 groups of 16 rows (nb <= 16) or 2 groups of 32: group g takes the quarters g, g + G, .. at the same
        // time.  Within a quarter the columns are still visited in ascending order by the lane that owns the row, so every sum
        // is the same sequence of operations as below: the same bits.
        const int shift = nb <= 16 ? 4 : 5;
        const int i = lane & ((1 << shift) - 1), G = 64 >> shift;
        const int cols = (nb + 3) >> 2;                                 // columns of the widest quarter
        for (int quarter = lane >> shift; quarter < 4; quarter += G) {
            const int jlo = (nb * quarter) >> 2, jhi = (nb * (quarter + 1)) >> 2;
            double ac = 0.0, aq = 0.0;
            for (int t = 0; t < cols; ++t) {
                const int j = jlo + t;
                if (j < jhi && i < nb) {
                    double c = C[i + (size_t)j * ld], q = Q[i + (size_t)j * ld];
                    f(i, j, c, q);
                    C[i + (size_t)j * ld] = c;
                    if (WRITE_Q) Q[i + (size_t)j * ld] = q;
                    const double kj = kvn[j];
                    ac += c * kj;
                    aq += q * kj;
                }
            }
            if (i < nb) {
                pnext[(quarter * 2 + 0) * lv + i] = ac;
                pnext[(quarter * 2 + 1) * lv + i] = aq;
            }
        }
        return;
    }
    for (int quarter = threadIdx.x >> 6; quarter < 4; quarter += SP_NTH >> 6) {     // one quarter per wave, or all four in turn
    const int jlo = (nb * quarter) >> 2, jhi = (nb * (quarter + 1)) >> 2;
    for (int i = lane; i < nb; i += 64) {
        double ac = 0.0, aq = 0.0;
        for (int j0 = jlo; j0 < jhi; j0 += RB) {
            double c[RB], q[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = (j0 + u < jhi) ? j0 + u : jhi - 1;      // clamped: the loads are unconditional and come first
                c[u] = C[i + (size_t)j * ld];
                q[u] = Q[i + (size_t)j * ld];
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = j0 + u;
                if (j < jhi) {
                    f(i, j, c[u], q[u]);
                    C[i + (size_t)j * ld] = c[u];
                    if (WRITE_Q) Q[i + (size_t)j * ld] = q[u];          // the sparse update leaves Q alone: read for Q k' only
                    const double kj = kvn[j];
                    ac += c[u] * kj;
                    aq += q[u] * kj;
                }
            }
        }
        pnext[(quarter * 2 + 0) * lv + i] = ac;
        pnext[(quarter * 2 + 1) * lv + i] = aq;
    }
    }
}

// ---- TRIANGULAR passes (the two- and four-wave shapes of the regular kernel, i.e. capacity > 64, Gaussian noise: SpAddParams::tri) ----------------------------------------
// C and Q are symmetric, and with a large basis the add path is bound by their stream: one read + one write of both per point
// (32 b^2 bytes).  In this mode the kernel works on the LOWER triangles only -- element (i, j), i >= j, at [i + j ld]; the upper
// triangle is ignored on entry and mirrored from the lower one when the patch leaves the kernel, so every other kernel still sees
// full matrices -- 16 b^2 bytes per point.  The element updates are the ones of the full passes, applied to the lower elements
// (same operations, same bits per element from the same inputs); what changes is that a mat-vec takes C_ji for C_ij above the
// diagonal (the full matrices differ from their transposes by a rounding of the rank-one terms) and sums a row in a different
// order: row part (j <= i, in the lane that owns row i) + column part (rows below the diagonal, summed over the lanes).  The
// results are therefore NOT the bits of the full mode (GPC_SPARSE_FULL=1 keeps it: the bit-identity tests between kernel shapes
// run there); they are gated like every other summation order by the parity statistics of tests/sparse_parity.py.
#define SP_TB 8     // columns per block of a triangular pass (8 C + 8 Q column sums = one 16-value wave reduction)
__device__ static __forceinline__ size_t sp_tri_at(int i, int j, int ld)
{
    return i >= j ? (size_t)i + (size_t)j * ld : (size_t)j + (size_t)i * ld;
}
// PACKED lower triangle (round 4: the state of a patch resident in LDS, sparse_add_kernel<.., RES>): column j holds its rows j .. ld - 1
// back to back, so element (i, j), i >= j, sits at  j ld - j (j - 1) / 2 + (i - j);  ld (ld + 1) / 2 doubles per matrix -- 41 KB at the
// reference's default capacity of 100, both matrices in half of a CU's LDS.  PK = false: the strided layout of the state in HBM.
template <bool PK>
__device__ static __forceinline__ size_t sp_at(int i, int j, int ld)        // i >= j
{
    if (PK) return (size_t)(j * ld - ((j * (j - 1)) >> 1) + (i - j));
    return (size_t)i + (size_t)j * ld;
}
template <bool PK>
__device__ static __forceinline__ size_t sp_sym_at(int i, int j, int ld)    // element (i, j) of the symmetric matrix out of its lower triangle
{
    return i >= j ? sp_at<PK>(i, j, ld) : sp_at<PK>(j, i, ld);
}
// first column of wave q's share (of nw = 4, 2 or 1 waves): equal areas of the lower triangle, 1 - sqrt(1 - q / nw), in multiples of SP_TB
__device__ static __forceinline__ int sp_tri_bound(int nb, int q, int nw)
{
    if (q <= 0) return 0;
    if (q >= nw) return nb;
    const double f = nw == 2 ? 0.2928932 : q == 1 ? 0.1339746 : q == 2 ? 0.2928932 : 0.5;
    const int c = ((int)(f * nb + 0.5 * SP_TB)) & ~(SP_TB - 1);
    return c < nb ? c : nb;
}
// Sums the four components of x over the 16 lanes of a DPP row: after the xor-8 and half-mirror rounds a lane keeps ONE component,
// sel = 2 (l >> 3 & 1) + (l >> 2 & 1), which the two quad rounds finish; lanes l & 15 = 0, 4, 8, 12 hold the totals of x0 .. x3.
__device__ static __forceinline__ double sp_row_reduce4(double x0, double x1, double x2, double x3, int r)
{
    const bool hi8 = (r & 8) != 0, hi4 = (r & 4) != 0;
    double k0 = hi8 ? x2 : x0, k1 = hi8 ? x3 : x1;
    const double s0 = hi8 ? x0 : x2, s1 = hi8 ? x1 : x3;
    k0 += sp_dpp<0x128>(s0);            // row_ror:8  (l <-> l ^ 8)
    k1 += sp_dpp<0x128>(s1);
    double k = hi4 ? k1 : k0;
    const double sd = hi4 ? k0 : k1;
    k += sp_dpp<0x141>(sd);             // row_half_mirror
    k += sp_dpp<0xB1>(k);               // quad_perm [1,0,3,2]
    k += sp_dpp<0x4E>(k);               // quad_perm [2,3,0,1]
    return k;
}
// One pass over the lower triangles (256 or 128 threads).  Wave w owns the columns [bound(w), bound(w + 1)); a wave instruction covers
// 16 rows x 4 columns -- lane l: row r = l & 15 of a 16-row group, column g = l >> 4 (and g + 4) of an SP_TB-column block -- so every
// 16-lane segment is one aligned cache line of one column, and the column sums of a block finish inside the DPP rows (no cross-row
// step: 45 instructions per block against 130 for the 16-value wave reduction of the first form, 64 rows x 1 column per instruction).
// A quad's row groups above the diagonal or beyond the basis are issued masked (584 instruction slots per pass at b = 200 for 314
// slots' worth of elements; skipping them behind wave-uniform guards was measured and not kept: the guards end the batch of loads).  Rows OUTSIDE, in quads of row groups, columns inside: the
// row parts of a quad are four register pairs whatever the basis size (with the columns outside they were 2 x 16 accumulators across
// an unrolled body: 110-145 spilled VGPRs), the column parts are reduced per (quad, block) and added up in LDS by the lane that owns
// the column -- the same lane every time, in program order: deterministic.  All 8 + 8 loads of a trip come first.
// MODE bits: 1 apply f(i, j, c, q) and store C | 2 store Q too | 4 mat-vecs with kv: row parts to prow [4][2][lv], column parts
// to pcol [2][lv] | 8 Q is neither read nor written.
template <int MODE, bool PK = false, class F>
__device__ static inline void sp_tri_pass(double* C, double* Q, int ld, int lv, int nb, const double* kv, double* prow, double* pcol, F f)
{
    constexpr bool UPD = (MODE & 1) != 0, WQ = (MODE & 2) != 0, MV = (MODE & 4) != 0, NOQ = (MODE & 8) != 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int nw = SP_NTH >> 6;
    const int jlo = sp_tri_bound(nb, w, nw), jhi = sp_tri_bound(nb, w + 1, nw);
    if (MV) {
        for (int j = jlo + lane; j < jhi; j += 64) pcol[j] = pcol[lv + j] = 0.0;
        if (nw < 4 && w == 0) {                        // two-wave shape: the row parts of the absent waves (the consumer adds four)
            for (int w2 = nw; w2 < 4; ++w2)
                for (int i = lane; i < nb; i += 64) prow[(w2 * 2 + 0) * lv + i] = prow[(w2 * 2 + 1) * lv + i] = 0.0;
        }
    }
    for (int Rq = 0; 16 * Rq < nb; Rq += 4) {
        double rc[4] = {0.0, 0.0, 0.0, 0.0}, rq[4] = {0.0, 0.0, 0.0, 0.0};     // row parts: rows 16 (Rq + t) + r over this lane's columns
        int ic[4];
        double kiv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 16 * (Rq + t) + r;
            ic[t] = i < nb ? i : nb - 1;
            if (MV) kiv[t] = kv[ic[t]];
        }
        const int jend = jhi < 16 * (Rq + 4) ? jhi : 16 * (Rq + 4);
        for (int j0 = jlo; j0 < jend; j0 += SP_TB) {
            const int ja = j0 + g, jb = j0 + g + 4;
            const bool oka = ja < jhi, okb = jb < jhi;
            const int jac = oka ? ja : jhi - 1, jbc = okb ? jb : jhi - 1;      // clamped: the loads are unconditional
            double ka = 0.0, kb = 0.0;
            if (MV) {
                ka = kv[jac];
                kb = kv[jbc];
            }
            double c[8], q[8];
            int at[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const int jc = s_ ? jbc : jac;
                    at[2 * t + s_] = (int)sp_at<PK>(ic[t] < jc ? jc : ic[t], jc, ld);
                    c[2 * t + s_] = C[at[2 * t + s_]];
                    q[2 * t + s_] = NOQ ? 0.0 : Q[at[2 * t + s_]];
                }
            }
            double cs0 = 0.0, cs1 = 0.0, cq0 = 0.0, cq1 = 0.0;                 // column parts of columns ja, jb: this lane's rows of the quad
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = 16 * (Rq + t) + r;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const int e = 2 * t + s_, j = s_ ? jb : ja;
                    if ((s_ ? okb : oka) && i >= j && i < nb) {
                        if (UPD) {
                            f(i, j, c[e], q[e]);
                            C[at[e]] = c[e];
                            if (WQ) Q[at[e]] = q[e];
                        }
                        if (MV) {
                            const double kj = s_ ? kb : ka;
                            rc[t] += c[e] * kj;
                            rq[t] += q[e] * kj;
                            if (i > j) {
                                if (s_) { cs1 += c[e] * kiv[t]; cq1 += q[e] * kiv[t]; }
                                else { cs0 += c[e] * kiv[t]; cq0 += q[e] * kiv[t]; }
                            }
                        }
                    }
                }
            }
            if (MV) {
                const double tot = sp_row_reduce4(cs0, cs1, cq0, cq1, r);      // r = 0, 4, 8, 12: C of ja, C of jb, Q of ja, Q of jb
                const int comp = r >> 2, j = (comp & 1) ? jb : ja;
                if ((r & 3) == 0 && j < jhi) pcol[(comp >> 1) * lv + j] += tot;
            }
        }
        if (MV) {
            // a row's four column groups
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double vc = rc[t], vq = rq[t];
                vc += __shfl_xor(vc, 16, 64);
                vq += __shfl_xor(vq, 16, 64);
                vc += __shfl_xor(vc, 32, 64);
                vq += __shfl_xor(vq, 32, 64);
                const int i = 16 * (Rq + t) + r;
                if (g == 0 && i < nb) {
                    prow[(w * 2 + 0) * lv + i] = vc;
                    prow[(w * 2 + 1) * lv + i] = vq;
                }
            }
        }
    }
}

// delete_bv(loc): sparse_gp.hpp:252-295 / sparse_gp_field.hpp:219-263.  b is workgroup-uniform; returns b-1.
template <int RB, bool TRI = false, bool PK = false>
__device__ static int sp_delete_bv(const SpState& S, int b, int loc, int field_bug, double* Cstar, double* Qstar,
                                   double* Crep, double* Qrep, bool tri_p = false)
{
    const bool tri = TRI && tri_p;      // (this patch runs in the triangular mode)
    const int tid = threadIdx.x, ld = S.ld, ldm = S.ldm, last = b - 1, ny = S.ny;
    double alphastar[3];
    for (int c = 0; c < ny; ++c) alphastar[c] = S.alpha[c * ld + loc];
    const double cstar = S.C[sp_at<PK>(loc, loc, ldm)];
    const double qstar = S.Q[sp_at<PK>(loc, loc, ldm)];
    for (int i = tid; i < b; i += SP_NTH) {
        if (tri) {                                   // columns of the symmetric matrices out of their lower triangles
            const size_t al = sp_sym_at<PK>(i, loc, ldm), aa = sp_sym_at<PK>(last, i, ldm);
            Cstar[i] = S.C[al];
            Qstar[i] = S.Q[al];
            Crep[i] = S.C[aa];
            Qrep[i] = S.Q[aa];
        } else {
            Cstar[i] = S.C[i + (size_t)loc * ldm];
            Qstar[i] = S.Q[i + (size_t)loc * ldm];
            Crep[i] = S.C[i + (size_t)last * ldm];
            Qrep[i] = S.Q[i + (size_t)last * ldm];
        }
    }
    __syncthreads();
    if (tid == 0) {
        Cstar[loc] = Cstar[last];   // Cstar(loc) = Cstar(last)  (:263)
        Qstar[loc] = Qstar[last];   // (:275)
        Crep[loc] = Crep[last];     // (:267)
        Qrep[loc] = Qrep[last];     // (:278)
    }
    __syncthreads();
    for (int i = tid; i < b; i += SP_NTH) {
        const double cr = Crep[i], qr = Qrep[i];
        if (tri) {
            const size_t al = sp_sym_at<PK>(i, loc, ldm);
            S.C[al] = cr;
            S.Q[al] = qr;
        } else {
            S.C[loc + (size_t)i * ldm] = cr;   // C.row(loc) = Crep^T
            S.C[i + (size_t)loc * ldm] = cr;   // C.col(loc) = Crep
            S.Q[loc + (size_t)i * ldm] = qr;
            S.Q[i + (size_t)loc * ldm] = qr;
        }
    }
    if (tid < ny) S.alpha[tid * ld + loc] = S.alpha[tid * ld + last];     // alpha(loc) = alpha(last)  (:257)
    if (tid == 32) {
        S.BV[2 * loc] = S.BV[2 * last];                                     // BV.col(loc) = BV.col(last) (:291)
        S.BV[2 * loc + 1] = S.BV[2 * last + 1];
    }
    __syncthreads();
    const int nb = b - 1;
    const double qc_den = qstar + cstar;
    // alpha update (:285) / field variant (:250-253, multiplies when bug-compatible)
    for (int i = tid; i < nb; i += SP_NTH) {
        const double qc = Qstar[i] + Cstar[i];
        for (int c = 0; c < ny; ++c) {
            if (ny == 1) S.alpha[i] -= alphastar[0] / qc_den * qc;
            else S.alpha[c * ld + i] -= alphastar[c] * (field_bug ? qc_den * qc : qc / qc_den);
        }
    }
    // C += Qs Qs^T / qstar - (Qs+Cs)(Qs+Cs)^T / (qstar+cstar);  Q -= Qs Qs^T / qstar   (:286-288)
    auto downdate = [&](int i, int j, double& c, double& q) {
        const double qq = (Qstar[i] * Qstar[j]) / qstar;
        const double cc = ((Qstar[i] + Cstar[i]) * (Qstar[j] + Cstar[j])) / qc_den;
        c += qq - cc;
        q -= qq;
    };
    if (tri) sp_tri_pass<1 | 2, PK>(S.C, S.Q, ldm, ld, nb, nullptr, nullptr, nullptr, downdate);
    else sp_rmw_cq<RB>(S.C, S.Q, ldm, nb, downdate);
    __syncthreads();
    return nb;
}

// Full update (sparse_gp.hpp:164-203) FUSED with the capacity deletion that must follow it when the basis is full
// (:206-223 -> delete_bv :252-295): the rank-1 growth C' = [C 0; 0 0] + r s s^T, Q' = [Q 0; 0 0] + e e^T / gamma and the
// rank-2 downdate of delete_bv touch every element of C and Q; run back to back they read and write both matrices twice
// (3.2 MB per point at b = 200, and the add path is HBM-bound).  Here the deletion candidate is scored on the updated
// diagonals, the two columns delete_bv needs (loc and the new last one) are formed from the OLD matrices plus the
// rank-1 terms, and one pass writes the final C and Q.  Every element goes through the same floating-point operations
// in the same order as the two-pass form, so the result is the same to the last bit; only the intermediate (b+1) x (b+1)
// matrices never reach memory.  b == capacity on entry and on return.
// When the coordinates of the NEXT point are known (nxt != nullptr) the pass also forms that point's mat-vecs (sp_rmw_cq_next).
template <int RB, bool TRI = false, bool PK = false>
__device__ static int sp_full_update_delete(const SpState& S, int b, double rr, double gamma, const double* qv, double px0, double px1,
                                            int field_bug, const double* ck, double* eh, double* sv, double* Cstar, double* Qstar,
                                            double* Crep, double* Qrep, double* anew, double* sval, int* sidx,
                                            const double* nxt, double* kvn, double* pnext, double sf, double c_exp, const double* T,
                                            double* pnext_col = nullptr, bool tri_p = false)
{
    const bool tri = TRI && tri_p;
    const int tid = threadIdx.x, ld = S.ld, ldm = S.ldm, ny = S.ny, last = b;
    const double ig = (double)1.0f / gamma;
    for (int i = tid; i <= b; i += SP_NTH) {
        const double si = (i < b) ? ck[i] : (double)1.0f;
        sv[i] = si;
        if (i == b) eh[b] = (double)(-1.0f);
        for (int c = 0; c < ny; ++c) {
            const double a0 = (i < b) ? S.alpha[c * ld + i] : 0.0;
            anew[c * ld + i] = a0 + qv[c] * si;
        }
    }
    __syncthreads();
    // capacity deletion candidate on the updated state (:206-223): argmin |alpha_i|^2 / (Q_ii + C_ii), first index wins ties
    double best = 0.0;
    int loc = 0x7fffffff;
    bool have = false;
    for (int i = tid; i <= b; i += SP_NTH) {
        double a2 = 0.0;
        for (int c = 0; c < ny; ++c) { const double a = anew[c * ld + i]; a2 += a * a; }
        const double c0 = (i < b) ? S.C[sp_at<PK>(i, i, ldm)] : 0.0, q0 = (i < b) ? S.Q[sp_at<PK>(i, i, ldm)] : 0.0;
        const double cd = c0 + (rr * sv[i]) * sv[i], qd = q0 + (ig * eh[i]) * eh[i];
        const double score = a2 / (qd + cd);
        if (!have || score < best) { best = score; loc = i; have = true; }
    }
    if (!have) best = __builtin_inf();
    sp_block_argmin(best, loc, sval, sidx);
    if (loc < 0 || loc > b) loc = 0;          // all-NaN scores: the reference keeps minloc = 0
    // columns loc and last of the updated matrices (delete_bv :259-278)
    for (int i = tid; i <= b; i += SP_NTH) {
        const bool old = (i < b) && (loc < b);
        const size_t al = tri ? sp_sym_at<PK>(old ? i : 0, old ? loc : 0, ldm) : (size_t)i + (size_t)loc * ldm;
        const double c0 = old ? S.C[al] : 0.0, q0 = old ? S.Q[al] : 0.0;
        Cstar[i] = c0 + (rr * sv[i]) * sv[loc];
        Qstar[i] = q0 + (ig * eh[i]) * eh[loc];
        Crep[i] = 0.0 + (rr * sv[i]) * sv[b];
        Qrep[i] = 0.0 + (ig * eh[i]) * eh[b];
    }
    __syncthreads();
    const double cstar = Cstar[loc], qstar = Qstar[loc];
    double alphastar[3];
    for (int c = 0; c < ny; ++c) alphastar[c] = anew[c * ld + loc];
    __syncthreads();
    if (tid == 0) {
        Cstar[loc] = Cstar[last];   // (:263)
        Qstar[loc] = Qstar[last];   // (:275)
        Crep[loc] = Crep[last];     // (:267)
        Qrep[loc] = Qrep[last];     // (:278)
    }
    if (tid == 32 && loc != last) {
        S.BV[2 * loc] = px0;        // BV.col(loc) = BV.col(last) = the new point (:291)
        S.BV[2 * loc + 1] = px1;
    }
    __syncthreads();
    const int nb = b;
    const double qc_den = qstar + cstar;
    // alpha (:257, :285; field variant :250-253 multiplies when bug-compatible)
    for (int i = tid; i < nb; i += SP_NTH) {
        const double qc = Qstar[i] + Cstar[i];
        for (int c = 0; c < ny; ++c) {
            const double asw = (i == loc) ? anew[c * ld + last] : anew[c * ld + i];
            if (ny == 1) S.alpha[i] = asw - alphastar[0] / qc_den * qc;
            else S.alpha[c * ld + i] = asw - alphastar[c] * (field_bug ? qc_den * qc : qc / qc_den);
        }
    }
    // the one pass over C and Q
    auto element = [&](int i, int j, double& c, double& q) {
        double bc, bq;
        if (i == loc) { bc = Crep[j]; bq = Qrep[j]; }
        else if (j == loc) { bc = Crep[i]; bq = Qrep[i]; }
        else {
            bc = c + (rr * sv[i]) * sv[j];
            bq = q + (ig * eh[i]) * eh[j];
        }
        const double qq = (Qstar[i] * Qstar[j]) / qstar;
        const double cc = ((Qstar[i] + Cstar[i]) * (Qstar[j] + Cstar[j])) / qc_den;
        c = bc + (qq - cc);
        q = bq - qq;
    };
    if (nxt) {
        // k' against the updated basis (BV[loc] was replaced above, behind a barrier)
        const double n0 = nxt[0], n1 = nxt[1];
        for (int i = tid; i < nb; i += SP_NTH) kvn[i] = gpc_rbf_neg(sf, c_exp, n0, n1, S.BV[2 * i], S.BV[2 * i + 1], T);
        __syncthreads();
        if (tri) sp_tri_pass<1 | 2 | 4, PK>(S.C, S.Q, ldm, ld, nb, kvn, pnext, pnext_col, element);
        else sp_rmw_cq_next<true, RB>(S.C, S.Q, ldm, ld, nb, kvn, pnext, element);
    } else {
        if (tri) sp_tri_pass<1 | 2, PK>(S.C, S.Q, ldm, ld, nb, nullptr, nullptr, nullptr, element);
        else sp_rmw_cq<RB>(S.C, S.Q, ldm, nb, element);
    }
    __syncthreads();
    return nb;
}

struct SpAddParams {
    gpc_params prm;
    double c_exp;
    int P, ny, ld, n_total;
    const int32_t* off;
    const double *x0, *x1, *y;
    const int32_t* perm;
    double *alpha, *C, *Q, *BV;
    int32_t *b, *count, *stat, *status_out;
    int fuse_next;   // 1: full-update passes also form the next point's mat-vecs (0 only through GPC_SPARSE_NO_FUSE, for the tests)
    const int32_t* start_it;   // per patch: points of this call already consumed by the small-basis kernel (nullptr: 0)
    int32_t* done_it;          // small-basis kernel only: how many points of this call it consumed
    uint8_t* trace;            // diagnostic: one decision byte per point of the call, in insertion order (or nullptr)
    // work list of the phases after the rows phase: the patches it did not finish, in the order they were handed over.  The kernels
    // that follow take list entries by ticket (one atomic per patch) instead of a static share of all P patches: only a fifth of the
    // patches reach them, and with static shares the wave that happens to own eight of those decides the launch time.
    int32_t* list;             // [P] patch ids (nullptr: static shares)
    int32_t* list_n;           // [0] entries in `list`, [1] .. [3] ticket counters of the launches that draw from it
    int32_t* out_list;         // where a phase that draws from `list` puts what it hands on (round 4: every later phase walks only the
    int32_t* out_list_n;       // entries it has work in, not the first phase's whole list: a ticket costs ~23 ns device-wide), or nullptr
    int ticket_slot;           // which counter this launch draws from
    int tri_min;               // triangular mode (sparse_add_kernel<false, ., true>): for patches that arrive with at least this many basis vectors
};

// Small-basis phase.  With the reference's default hyper-parameters the basis stays at a dozen vectors whatever the capacity,
// and a point is a handful of short loops whose cost is the latency of C and Q in L2 / HBM.  sparse_add_kernel<true> runs one
// wave per patch with C and Q (up to SP_BMAX x SP_BMAX) resident in LDS, ten patches per CU, and hands a patch over -- state written back, the
// number of points consumed recorded in done_it -- at the first point that would grow its basis beyond SP_BMAX; the regular
// kernel then continues from there (start_it).  Same operations in the same order: the two-phase run leaves the states of
// a one-phase run, bit for bit.

// PROBIT: the probit functor (sqrt, erf, exp: ~450 instructions and their constants) is compiled in only where it is used -- in the
// Gaussian instantiation, the reference's production path, its registers go to the point loop
// TRI: the triangular mode of the four-wave shape (see sp_tri_pass)
// RES (round 4; implies TRI): the lower triangles of C and Q of the patch in flight are RESIDENT IN LDS, packed (sp_at<true>), loaded once
// when the patch enters the kernel and written back (both triangles) once when it leaves -- the add call moves 16 b^2 bytes per patch instead
// of 16 b^2 per POINT.  Fits while 2 x (capacity + 1)(capacity + 2) / 2 doubles + the vectors stay within the CU's 160 KB: capacity <= 120,
// which covers the reference's default of 100 (/root/reference/src/sparse_gp.h:48).  One workgroup of four waves per CU.  Every patch runs
// the triangular passes (whatever basis it arrives with); same element updates, same summation order as the HBM-resident triangular mode.
template <bool SMALL, bool PROBIT = false, bool TRI = false, bool RES = false, int BM = SP_BMAX>
__global__ __launch_bounds__(SMALL ? 64 : SP_THREADS, SMALL ? 4 : 2) void sparse_add_kernel(SpAddParams A)   // <= 256 VGPRs (128 for the small-basis phase)
{
    static_assert(!RES || (TRI && !SMALL && !PROBIT), "the LDS-resident mode is a form of the triangular mode");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ldg = A.ld, ny = A.ny;               // ldg: strides of the state in global memory
    const int ld = SMALL ? BM + 1 : ldg;      // stride of the LDS vectors (and of alpha / BV while they live in LDS)
    const int ldm = SMALL ? BM : ldg;         // stride of C and Q where the update loops see them
    constexpr int RB = SMALL ? 2 : SP_RMW;         // elements per thread and trip of the update passes (LDS needs no deep batches)
    double* T = reinterpret_cast<double*>(smem);   // 64
    double* red = T + 64;                          // 16
    double* sval = red + 16;                       // 4
    int* sidx = reinterpret_cast<int*>(sval + 4);  // 4 ints (2 doubles)
    double* kv = sval + 6;                         // k        [ld]
    double* ck = kv + ld;                          // C k      [ld]
    double* eh = ck + ld;                          // e_hat    [ld+1]
    double* sv = eh + ld + 1;                      // s / s_hat[ld+1]
    double* part = sv + ld + 1;                    // partial mat-vec sums [4][2][ld]
    double* Cstar = part;                          // delete_bv scratch aliases the mat-vec partials
    double* Qstar = part + ld;
    double* Crep = part + 2 * ld;
    double* Qrep = part + 3 * ld;
    double* pnext = part + 8 * ld;                 // the NEXT point's mat-vec partials, formed inside the fused update pass
    double* kvn = pnext + 8 * ld;                  // ... and its k  [ld]
    double* alphaL = kvn + ld;                     // alpha [ny][ld] and BV [ld][2] of the patch in flight: read and updated at every
    double* BVL = alphaL + 3 * ld;                 // point, so they live here between the first and the last point of the call
    double* Cl = BVL + 2 * ld;                     // SMALL: C, Q [BM][BM]
    double* Ql = Cl + BM * BM;
    double* part_col = BVL + 2 * ld;               // TRI: column parts of the mat-vecs [2][ld], this point's and the next one's
    double* pnext_col = part_col + 2 * ld;
    const int ldp = A.prm.capacity + 1;            // RES: leading dimension of the packed triangles
    double* Cp = pnext_col + 2 * ld;               // RES: C, Q lower triangles, packed [ldp (ldp + 1) / 2]
    double* Qp = Cp + (ldp * (ldp + 1)) / 2;
    gpc_exp_table_init(T);

    const double sf = A.prm.sigmaf_sq, s20 = A.prm.noise, eps_tol = A.prm.eps_tol;
    const int capacity = A.prm.capacity;

    int* s_ticket = reinterpret_cast<int*>(red + 15);             // (the last slot of the reduction scratch is never used: <= 3 sums x 4 waves)
    for (int slot = blockIdx.x;; slot += gridDim.x) {
        int patch = slot;
        if (A.list) {                                                // work list: the next entry nobody has taken yet
            // (the first entry of a workgroup is its own index: tickets to ONE counter are served at ~23 ns each device-wide -- the
            // 32768 skipped entries of a C4-fill call take 0.76 ms that way -- and at launch every workgroup asks at once)
            int idx = slot;
            if (slot != (int)blockIdx.x) {
                __syncthreads();
                if (tid == 0) *s_ticket = (int)gridDim.x + atomicAdd(A.list_n + A.ticket_slot, 1);
                __syncthreads();
                idx = *s_ticket;
            }
            if (idx >= A.list_n[0]) break;
            patch = A.list[idx];
        } else if (patch >= A.P) {
            break;
        }
        const int o = A.off[patch], n = A.off[patch + 1] - o;
        SpState S;
        S.ld = ld; S.ny = ny; S.ldm = RES ? ldp : ldm;
        double* const alphag = A.alpha + (size_t)patch * ny * ldg;
        double* const BVg = A.BV + (size_t)patch * ldg * 2;
        double* const Cg = A.C + (size_t)patch * ldg * ldg;
        double* const Qg = A.Q + (size_t)patch * ldg * ldg;
        S.alpha = alphaL;
        S.C = RES ? Cp : SMALL ? Cl : Cg;
        S.Q = RES ? Qp : SMALL ? Ql : Qg;
        S.BV = BVL;
        int b = A.b[patch];
        int st = A.stat[patch];
        // the triangular passes pay from a mid-sized basis on (their per-block reductions and the uneven shares of the four waves cost
        // more than half a stream of a small matrix saves): decided per patch and call from the basis it arrives with
        [[maybe_unused]] const bool tri = RES || (TRI && b >= A.tri_min);
        const int it0 = A.start_it ? A.start_it[patch] : 0;     // an earlier phase (rows / small-basis) already took these
        __syncthreads();
        if (SMALL && (b > BM || n == 0)) {          // too large from the start (or nothing to do): all of it is the regular kernel's
            if (tid == 0) {
                A.done_it[patch] = it0;
                if (A.out_list && n > 0 && it0 < n) A.out_list[atomicAdd(A.out_list_n, 1)] = patch;
            }
            continue;
        }
        if (A.start_it && n > 0 && it0 >= n) continue;   // an earlier phase finished this patch (and wrote its state, status and done_it)
        for (int i = tid; i < b; i += SP_NTH) {
            BVL[2 * i] = BVg[2 * i];
            BVL[2 * i + 1] = BVg[2 * i + 1];
            for (int c = 0; c < ny; ++c) alphaL[c * ld + i] = alphag[c * ldg + i];
        }
        if (SMALL) {
            for (int e = tid; e < b * b; e += SP_NTH) {
                const int i = e % b, j = e / b;
                Cl[i + j * BM] = Cg[i + (size_t)j * ldg];
                Ql[i + j * BM] = Qg[i + (size_t)j * ldg];
            }
        }
        if (RES) {                                       // the lower triangles come on chip (the upper ones are redundant: every producer mirrors)
            for (int e = tid; e < b * b; e += SP_NTH) {
                const int i = e % b, j = e / b;
                if (i >= j) {
                    Cp[sp_at<true>(i, j, ldp)] = Cg[i + (size_t)j * ldg];
                    Qp[sp_at<true>(i, j, ldp)] = Qg[i + (size_t)j * ldg];
                }
            }
        }
        int it_end = n;                                  // SMALL: where this phase stopped
        __syncthreads();

        bool have_next = false;                      // k and the mat-vec partials of this iteration's point were formed by the previous one
        // The coordinates and targets of a point sit behind two dependent global loads (insertion order, then the point): they
        // are fetched one point ahead, and the index two ahead, so that nothing of a point waits on them (in the small-basis
        // regime those round trips were a quarter of a point).
        double cx0 = 0.0, cx1 = 0.0, cy[3] = {0.0, 0.0, 0.0};       // the current point
        int r_nxt = 0;                                                // row of point it + 1
        if (it0 < n) {
            const int r0 = A.perm ? A.perm[o + it0] : it0;
            cx0 = A.x0[o + r0];
            cx1 = A.x1[o + r0];
            for (int c = 0; c < ny; ++c) cy[c] = A.y[(size_t)c * A.n_total + o + r0];
            if (it0 + 1 < n) r_nxt = A.perm ? A.perm[o + it0 + 1] : it0 + 1;
        }
        double nx0 = 0.0, nx1 = 0.0, nyv[3] = {0.0, 0.0, 0.0};      // point it + 1, in flight
        int r_nxt2 = 0;
        for (int it = it0; it < n; ++it, cx0 = nx0, cx1 = nx1, cy[0] = nyv[0], cy[1] = nyv[1], cy[2] = nyv[2], r_nxt = r_nxt2) {
            const double px0 = cx0, px1 = cx1;
            double yv[3] = {cy[0], cy[1], cy[2]};
            const bool more_pts = it + 1 < n;
            if (more_pts) {
                nx0 = A.x0[o + r_nxt];
                nx1 = A.x1[o + r_nxt];
                for (int c = 0; c < ny; ++c) nyv[c] = A.y[(size_t)c * A.n_total + o + r_nxt];
                if (it + 2 < n) r_nxt2 = A.perm ? A.perm[o + it + 2] : it + 2;
            }
            const double kstar = sf;   // kernel_function(X, X) = p(0)*exp(0)  (:98)

            const bool from_prev = have_next;
            have_next = false;
            if (b == 0) {
                // First point (:100-114)
                if (tid == 0) {
                    for (int c = 0; c < ny; ++c) S.alpha[c * ld] = yv[c] / (kstar + s20);
                    S.C[0] = (double)(-1.0f) / (kstar + s20);
                    S.Q[0] = (double)(1.0f) / kstar;
                    S.BV[0] = px0;
                    S.BV[1] = px1;
                }
                b = 1;
                if (A.trace && tid == 0) A.trace[o + it] = 0x81;
                __syncthreads();
                continue;
            }
            int dec = 0;     // decision byte (include/gpc.h, gpc_sparse_set_trace): bit 0 full update, bits 1-3 / 4-6 deletions

            // k = construct_covariance(X, BV)  (:119, :523-530)
            if (from_prev) {
                double* t_ = kv; kv = kvn; kvn = t_;       // formed with the update pass of the previous point
            } else {
                for (int i = tid; i < b; i += SP_NTH) kv[i] = gpc_rbf_neg(sf, A.c_exp, px0, px1, S.BV[2 * i], S.BV[2 * i + 1], T);
            }
            __syncthreads();

            // C k and e_hat = Q k (:140,:160,:171): wave w covers columns j in its quarter, lanes cover rows
            const double* pp = from_prev ? pnext : part;
            [[maybe_unused]] const double* pc = from_prev ? pnext_col : part_col;
            if (TRI && tri && !from_prev) {
                sp_tri_pass<4, RES>(S.C, S.Q, S.ldm, ld, b, kv, part, part_col, [](int, int, double&, double&) {});
            } else if (!from_prev && SMALL && b <= 32) {
                // one wave, small basis: the four quarters side by side (see sp_rmw_cq_next)
                const int shift = b <= 16 ? 4 : 5;
                const int i = lane & ((1 << shift) - 1), G = 64 >> shift;
                for (int quarter = lane >> shift; quarter < 4; quarter += G) {
                    const int jlo = (b * quarter) >> 2, jhi = (b * (quarter + 1)) >> 2;
                    double ac = 0.0, aq = 0.0;
                    if (i < b) {
                        for (int j = jlo; j < jhi; ++j) {
                            const double kj = kv[j];
                            ac += S.C[i + (size_t)j * ldm] * kj;
                            aq += S.Q[i + (size_t)j * ldm] * kj;
                        }
                        part[(quarter * 2 + 0) * ld + i] = ac;
                        part[(quarter * 2 + 1) * ld + i] = aq;
                    }
                }
            } else if (!from_prev) {
                for (int quarter = wave; quarter < 4; quarter += SP_NTH >> 6) {     // one quarter per wave, or all four in turn
                    const int jlo = (b * quarter) >> 2, jhi = (b * (quarter + 1)) >> 2;
                    for (int i = lane; i < b; i += 64) {
                        double ac = 0.0, aq = 0.0;
                        for (int j = jlo; j < jhi; ++j) {
                            const double kj = kv[j];
                            ac += S.C[i + (size_t)j * ldm] * kj;
                            aq += S.Q[i + (size_t)j * ldm] * kj;
                        }
                        part[(quarter * 2 + 0) * ld + i] = ac;
                        part[(quarter * 2 + 1) * ld + i] = aq;
                    }
                }
            }
            __syncthreads();
            double sums[4] = {0.0, 0.0, 0.0, 0.0};   // m[0..2] partial, (kCk, ke) handled in a second pass
            double dots[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = tid; i < b; i += SP_NTH) {
                double c_ = pp[0 * ld + i] + pp[2 * ld + i] + pp[4 * ld + i] + pp[6 * ld + i];
                double q_ = pp[1 * ld + i] + pp[3 * ld + i] + pp[5 * ld + i] + pp[7 * ld + i];
                if (TRI && tri) {                            // + the column part: rows below the diagonal
                    c_ += pc[i];
                    q_ += pc[ld + i];
                }
                ck[i] = c_;
                eh[i] = q_;
                const double ki = kv[i];
                dots[0] += ki * c_;                        // k^T C k   (:122)
                dots[1] += ki * q_;                        // k^T e_hat (:144)
                for (int c = 0; c < ny; ++c) sums[c] += S.alpha[c * ld + i] * ki;   // m = alpha^T k (:121)
            }
            sp_block_sum4<2>(dots, red);
            if (ny == 1) sp_block_sum4<1>(sums, red);
            else sp_block_sum4<3>(sums, red);
            const double s2 = kstar + dots[0];
            double gamma = kstar - dots[1];
            if (gamma < (double)1e-12f) gamma = 0;          // :146-151

            // r = noise.dx2_ln, q = noise.dx_ln  (/root/reference/src/gaussian_noise.cpp:9-18, gaussian_noise_3d.cpp:11-20,
            // probit_noise.cpp:11-31)
            double rr, qv[3];
            if (PROBIT && A.prm.noise_model != GPC_NOISE_GAUSSIAN && ny == 1) {
                if constexpr (PROBIT) gpc_probit_q_r(A.prm.noise_model, s20, yv[0], sums[0], s2, &qv[0], &rr);
            } else {
                rr = (double)(-1.0f) / (s20 + s2);
                for (int c = 0; c < ny; ++c) qv[c] = (yv[c] - sums[c]) / (s20 + s2);
            }

            if (gamma < eps_tol && capacity != -1) {
                // sparse update (:155-163)
                const double eta = 1 / (1 + gamma * rr);
                for (int i = tid; i < b; i += SP_NTH) {
                    const double sh = ck[i] + eh[i];        // s_hat = C*k + e_hat
                    sv[i] = sh;
                    for (int c = 0; c < ny; ++c) S.alpha[c * ld + i] += sh * (qv[c] * eta);
                }
                __syncthreads();
                const double re = rr * eta;
                if (A.fuse_next && it + 1 < n) {
                    // the basis does not change: the next point's k against it, and its mat-vecs out of this pass (Q is only read)
                    const double n0 = nx0, n1 = nx1;
                    for (int i = tid; i < b; i += SP_NTH) kvn[i] = gpc_rbf_neg(sf, A.c_exp, n0, n1, S.BV[2 * i], S.BV[2 * i + 1], T);
                    __syncthreads();
                    auto proj = [&](int i, int j, double& c, double&) { c = c + (re * sv[i]) * sv[j]; };
                    if (TRI && tri) sp_tri_pass<1 | 4, RES>(S.C, S.Q, S.ldm, ld, b, kvn, pnext, pnext_col, proj);
                    else sp_rmw_cq_next<false, RB>(S.C, S.Q, ldm, ld, b, kvn, pnext, proj);
                    have_next = true;
                } else if (TRI && tri) {
                    sp_tri_pass<1 | 8, RES>(S.C, S.Q, S.ldm, ld, b, nullptr, nullptr, nullptr, [&](int i, int j, double& c, double&) { c = c + (re * sv[i]) * sv[j]; });
                } else {
                    const int nn = b * b;
                    for (int e0 = tid; e0 < nn; e0 += SP_NTH * RB) {     // loads first, see sp_rmw_cq
                        double cv[RB];
                        int ii[RB], jj[RB];
#pragma unroll
                        for (int u = 0; u < RB; ++u) {
                            const int e = e0 + u * SP_NTH, ec = e < nn ? e : e0;
                            ii[u] = ec % b;
                            jj[u] = ec / b;
                            cv[u] = S.C[ii[u] + (size_t)jj[u] * ldm];
                        }
#pragma unroll
                        for (int u = 0; u < RB; ++u)
                            if (e0 + u * SP_NTH < nn) S.C[ii[u] + (size_t)jj[u] * ldm] = cv[u] + (re * sv[ii[u]]) * sv[jj[u]];
                    }
                }
                __syncthreads();
            } else if (SMALL && b + 1 > BM && !(capacity > 0 && capacity <= BM)) {
                it_end = it;                                // this point would grow the basis beyond the resident block: nothing of it
                break;                                      // has been applied yet -- the regular kernel redoes it from the state as it is
            } else if (b >= ldg) {
                st = GPC_STATUS_OVERFLOW;                   // capacity == -1 and GPC_MAX_BV reached: skip the point
            } else if (capacity > 0 && b + 1 > capacity) {
                // full update + the capacity deletion it forces, in one pass over C and Q
                double nx[2] = {0.0, 0.0};
                const bool more = A.fuse_next && it + 1 < n;
                if (more) {
                    nx[0] = nx0;
                    nx[1] = nx1;
                }
                b = sp_full_update_delete<RB, TRI, RES>(S, b, rr, gamma, qv, px0, px1, A.prm.ref_field_delete_bug, ck, eh, sv, Cstar, Qstar, Crep,
                                               Qrep, part + 4 * ld, sval, sidx, more ? nx : nullptr, kvn, pnext, sf, A.c_exp, T, pnext_col, tri);
                have_next = more;
                dec = 1 | 2;
            } else {
                // full update (:164-203)
                dec = 1;
                for (int i = tid; i <= b; i += SP_NTH) {
                    const double si = (i < b) ? ck[i] : (double)1.0f;
                    sv[i] = si;
                    if (i == b) eh[b] = (double)(-1.0f);
                    for (int c = 0; c < ny; ++c) {
                        const double a0 = (i < b) ? S.alpha[c * ld + i] : 0.0;
                        S.alpha[c * ld + i] = a0 + qv[c] * si;
                    }
                }
                if (tid == 64 % SP_NTH) {
                    S.BV[2 * b] = px0;
                    S.BV[2 * b + 1] = px1;
                }
                __syncthreads();
                const double ig = (double)1.0f / gamma;
                const int nb = b + 1;
                auto grow = [&](int i, int j, double& c, double& q) {
                    const bool old = (i < b) && (j < b);          // the new row / column starts from zero, whatever memory held
                    c = (old ? c : 0.0) + (rr * sv[i]) * sv[j];
                    q = (old ? q : 0.0) + (ig * eh[i]) * eh[j];
                };
                if (A.fuse_next && it + 1 < n) {
                    // the next point's k against the grown basis, and its mat-vecs out of this pass
                    const double n0 = nx0, n1 = nx1;
                    for (int i = tid; i < nb; i += SP_NTH) kvn[i] = gpc_rbf_neg(sf, A.c_exp, n0, n1, S.BV[2 * i], S.BV[2 * i + 1], T);
                    __syncthreads();
                    if (TRI && tri) sp_tri_pass<1 | 2 | 4, RES>(S.C, S.Q, S.ldm, ld, nb, kvn, pnext, pnext_col, grow);
                    else sp_rmw_cq_next<true, RB>(S.C, S.Q, ldm, ld, nb, kvn, pnext, grow);
                    have_next = true;
                } else {
                    if (TRI && tri) sp_tri_pass<1 | 2, RES>(S.C, S.Q, S.ldm, ld, nb, nullptr, nullptr, nullptr, grow);
                    else sp_rmw_cq<RB>(S.C, S.Q, ldm, nb, grow);
                }
                b = nb;
                __syncthreads();
            }

            // Delete BVs if necessary (:206-223)
            while (b > capacity && capacity > 0) {
                double best = 0.0;
                int loc = 0x7fffffff;
                bool have = false;
                for (int i = tid; i < b; i += SP_NTH) {
                    double a2 = 0.0;
                    for (int c = 0; c < ny; ++c) { const double a = S.alpha[c * ld + i]; a2 += a * a; }
                    const double score = a2 / (S.Q[sp_at<RES>(i, i, S.ldm)] + S.C[sp_at<RES>(i, i, S.ldm)]);
                    if (!have || score < best) { best = score; loc = i; have = true; }
                }
                if (!have) best = __builtin_inf();
                sp_block_argmin(best, loc, sval, sidx);
                if (loc < 0 || loc >= b) loc = 0;          // all-NaN scores: the reference keeps minloc = 0
                b = sp_delete_bv<RB, TRI, RES>(S, b, loc, A.prm.ref_field_delete_bug, Cstar, Qstar, Crep, Qrep, tri);
                have_next = false;
                if (((dec >> 1) & 7) < 7) dec += 2;
            }
            // Delete for geometric reasons (:226-242)
            {
                double minscore = 0.0;
                while (minscore < (double)1e-9f && b > 1) {
                    double best = 0.0;
                    int loc = 0x7fffffff;
                    bool have = false;
                    for (int i = tid; i < b; i += SP_NTH) {
                        const double score = (double)1.0f / S.Q[sp_at<RES>(i, i, S.ldm)];
                        if (!have || score < best) { best = score; loc = i; have = true; }
                    }
                    if (!have) best = __builtin_inf();
                    // one wave: if no lane holds a score below the threshold the loop ends whatever the minimum is (>= 1e-9f, or NaN:
                    // both leave it, nothing else reads minscore) -- the arg-min is only needed to find WHICH vector goes
                    if (SP_NTH == 64 && __builtin_amdgcn_ballot_w64(have && best < (double)1e-9f) == 0) break;
                    sp_block_argmin(best, loc, sval, sidx);
                    if (loc < 0 || loc >= b) loc = 0;
                    minscore = best;
                    if (minscore < (double)1e-9f) {
                        b = sp_delete_bv<RB, TRI, RES>(S, b, loc, A.prm.ref_field_delete_bug, Cstar, Qstar, Crep, Qrep, tri);
                        have_next = false;              // the matrices and the basis changed after the pass
                        if (((dec >> 4) & 7) < 7) dec += 16;
                    }
                    else if (!(minscore >= (double)1e-9f)) break;   // NaN: `minscore < 1e-9f` is false in the reference too
                }
            }
            // isnan(C(0,0)) -> "sparse_gp::C has become Nan" (:245)
            {
                const double c00 = S.C[0];
                if (c00 != c00 && st == GPC_STATUS_OK) st = GPC_STATUS_NAN;
            }
            if (A.trace && tid == 0) A.trace[o + it] = (uint8_t)dec;
        }
        __syncthreads();
        for (int i = tid; i < b; i += SP_NTH) {
            BVg[2 * i] = BVL[2 * i];
            BVg[2 * i + 1] = BVL[2 * i + 1];
            for (int c = 0; c < ny; ++c) alphag[c * ldg + i] = alphaL[c * ld + i];
        }
        if (SMALL) {
            for (int e = tid; e < b * b; e += SP_NTH) {
                const int i = e % b, j = e / b;
                Cg[i + (size_t)j * ldg] = Cl[i + j * BM];
                Qg[i + (size_t)j * ldg] = Ql[i + j * BM];
            }
            if (tid == 0) {
                A.done_it[patch] = it_end;
                if (A.out_list && it_end < n) A.out_list[atomicAdd(A.out_list_n, 1)] = patch;      // handed on: the next phase continues it
            }
        }
        if (RES) {
            // the state goes back to HBM, both triangles (every other kernel reads full matrices): 16 b^2 bytes once per call
            for (int e = tid; e < b * b; e += SP_NTH) {
                const int i = e % b, j = e / b;
                if (i >= j) {
                    const double cv_ = Cp[sp_at<true>(i, j, ldp)], qv_ = Qp[sp_at<true>(i, j, ldp)];
                    Cg[i + (size_t)j * ldg] = cv_;
                    Qg[i + (size_t)j * ldg] = qv_;
                    if (i > j) {
                        Cg[j + (size_t)i * ldg] = cv_;
                        Qg[j + (size_t)i * ldg] = qv_;
                    }
                }
            }
        } else if (TRI && tri) {
            // the upper triangles from the lower ones: every other kernel (and the full mode) reads full matrices.  16 x 16 tiles, one
            // per 16 lanes and trip, read along their columns (once per call: 16 b^2 bytes against 16 b^2 per POINT)
            const int nt_ = (b + 15) >> 4, sub = tid >> 4, r = tid & 15;
            for (int tl = sub; tl < nt_ * nt_; tl += SP_NTH >> 4) {
                const int ti = tl % nt_, tj = tl / nt_;
                if (ti < tj) continue;
                for (int c = 0; c < 16; ++c) {
                    const int i = 16 * ti + r, j = 16 * tj + c;
                    if (i < b && j < b && i > j) {
                        Cg[j + (size_t)i * ldg] = Cg[i + (size_t)j * ldg];
                        Qg[j + (size_t)i * ldg] = Qg[i + (size_t)j * ldg];
                    }
                }
            }
        }
        if (tid == 0) {
            A.b[patch] = b;
            A.count[patch] += it_end - it0;
            A.stat[patch] = st;
            if (A.status_out) A.status_out[patch] = st;
        }
    }
}

// ---- rows phase: SEVERAL PATCHES PER WAVE -----------------------------------------------------------------------------
// With the reference's default hyper-parameters (src/rbf_kernel.h:24, src/sparse_gp.h:48) a patch keeps about 13 basis vectors
// and 95 % of its points take the projected update (src/sparse_gp.hpp:155-163): one wave per patch leaves 50 of 64 lanes idle
// and the pass is bound by the instruction count of a point's serial skeleton.  Here a patch owns G = 16 (or 32) lanes -- a DPP
// row, where the reductions of a point already live (sp_wave_sum_dpp) -- so a wave carries 4 (2) patches through the same
// instruction stream.  Lane i of a patch holds row i of the basis (k_i, (C k)_i, e_hat_i, alpha_i, BV_i and -- round 4 -- its rows of
// C and Q in registers, in the SLOT layout described below); the B x B blocks in LDS are where a patch is loaded, grown and written
// back; the lanes of different patches diverge only where their branch decisions differ (predication).
// The phase covers what needs no deletion -- first point, projected updates, and full updates that keep the basis within B
// vectors -- and hands a patch over (state written back, points consumed in done_it) BEFORE the first point that needs anything
// else.  Round 4: a second instance (two patches per wave, 24 rows, the listed patches by ticket) continues from there, then the
// one-wave kernel with a resident block of 48 (deletions in place), then the regular kernel; each phase appends what it hands on to
// the next phase's list (gpc_sparse_add_dev).
// A geometric deletion (src/sparse_gp.hpp:226-242) can only follow a full update (the projected update leaves Q alone), and
// whether one would follow is decided from the updated diagonal before anything is written.
// Same operations in the same order per patch as sparse_add_kernel (quarter-wise mat-vec sums, DPP row sums, the next
// point's mat-vecs formed from the updated values): the states are the same bit for bit
// (tests/test_sparse_gpu.py::test_sparse_rows_phase_is_bit_identical).  Gaussian noise only.
template <int G>
__device__ static __forceinline__ double sp_row_sum(double v)
{
    v += sp_dpp<0xB1>(v);      // the four steps of sp_wave_sum_dpp inside the row of 16
    v += sp_dpp<0x4E>(v);
    v += sp_dpp<0x141>(v);
    v += sp_dpp<0x140>(v);
    if (G == 32) v += __shfl_xor(v, 16, 64);           // (row 0 + row 16): the first pair of sp_wave_sum_dpp
    // the absent rows of the one-patch-per-wave layout hold +0.0 terms: (R + 0.0) + (0.0 + 0.0), then the +0.0 of the one-wave shape
    return v + 0.0;
}

// SLOT layout of the rows phase (round 4).  The mat-vec sums of every kernel shape run over four QUARTERS of the columns, quarter q =
// columns [(b q) >> 2, (b (q + 1)) >> 2), each summed in column order -- that is what makes the shapes agree bit for bit.  Indexed by column,
// a lane pays for every slot (q, t) of its row an index, a bound check, a clamp and two addresses: ~7 of the ~10 VALU operations of the
// slot, in a kernel whose time is its VALU count.  So the rows phase keeps C, Q and the vectors k, s, e_hat, k_next by SLOT instead:
// column j of quarter q sits at physical column QN q + (j - start_q), and the slots a quarter does not fill hold +0.0 in every row and
// vector.  A pass is then 16 slots at compile-time addresses with no mask at all: an empty slot computes 0 + x * 0 = +0.0, stores it back
// and adds (+0.0) * (+0.0) to a sum that started at +0.0 -- the same bits as skipping it, as long as x is finite.  The layout moves only
// when the basis grows (columns shift LEFT, never onto a column that has not been read yet if they are taken in column order --
// tests/test_sparse_gpu.py checks the states bit for bit against the other shapes).
template <int QN>
__device__ static __forceinline__ int sp_slot(int j, int b)
{
    const int s1 = b >> 2, s2 = b >> 1, s3 = (3 * b) >> 2;
    const int q = (j >= s1 ? 1 : 0) + (j >= s2 ? 1 : 0) + (j >= s3 ? 1 : 0);
    const int s = q == 0 ? 0 : q == 1 ? s1 : q == 2 ? s2 : s3;
    return QN * q + (j - s);
}

#define SP_FOR_C(c) _Pragma("unroll") for (int c = 0; c < 3; ++c) if (c < ny)   /* static index: the planes stay in registers */
// G lanes per patch, B <= G rows / slots of state (B = G = 16: the first phase, four patches per wave; G = 32, B = 24 = SP_BMAX: the SECOND
// phase, two patches per wave, which takes the patches of the work list -- by ticket -- that the first phase handed over, until they
// outgrow 24 vectors: what the one-wave kernel sparse_add_kernel<true> used to do at ~560 VALU operations per point and patch).
template <int G, int NY, int B = G, bool LIST = false>
__global__ __launch_bounds__(64, 2) void sparse_add_rows_kernel(SpAddParams A)
{
    constexpr int R = 64 / G, QN = B / 4;
    static_assert(B % 4 == 0 && B <= G, "rows of state per patch");
    // A lane keeps its rows of C and Q in REGISTERS between full updates (the passes touch nothing else of the matrices, and
    // every slot index in them is a compile-time value): the update pass reads two vectors from LDS instead of two vectors and two
    // matrices -- 16 KB instead of 40 KB per wave and point through the CU's one LDS pipe, which was as busy as the VALUs.  The LDS
    // blocks stay the place where a patch is loaded, grown (the full update moves columns between slots) and written back; Q in LDS
    // is always current (only the full update writes it), C in LDS only after rows_to_lds().
    // (B == 24, the second phase: 96 registers of rows, and still fewer spilled ones than with the pass's operands in flight -- 176
    // against 256 B of scratch -- and 24 KB instead of 60 KB of LDS traffic per step: 4.90 -> 4.68 ms of add kernels per defaults pass)
    constexpr bool REG = (B == 16 && G == 16) || B == 24;
    constexpr int UNR = (G == 16 || B == 24) ? QN : 2;           // trips of the column loops unrolled together (registers; all of them: slot addresses are immediates)
    constexpr int ROWD = 2 * B * B + 4 * B + (B == G ? 16 : 0);      // doubles of LDS per patch row (+16: de-phases the rows' banks)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* T = reinterpret_cast<double*>(smem);   // 64
    const int lane = threadIdx.x, r = lane / G, i = lane % G;
    double* Cl = T + 64 + r * ROWD;                // C [B slots][B rows]: row i of the column in slot p at i + B p (see sp_slot)
    double* Ql = Cl + B * B;
    double* kvL = Ql + B * B;                      // k of the current point, by slot
    double* svL = kvL + B;                         // s / s_hat
    double* ehL = svL + B;                         // e_hat
    double* knL = ehL + B;                         // k of the next point
    gpc_exp_table_init(T);
    __syncthreads();
    const int ldg = A.ld;
    constexpr int ny = NY;                         // the channel count is a compile-time value here: with a run-time one the compiler
                                                   // evaluates all three channels and selects (6 .. 9 VALU operations per channel loop)
    const double sf = A.prm.sigmaf_sq, s20 = A.prm.noise, eps_tol = A.prm.eps_tol;
    const int capacity = A.prm.capacity;
    const double kstar = sf;

    for (int base = blockIdx.x * R;; base += gridDim.x * R) {
        int patch;
        bool valid;
        if (LIST) {
            // the patches of the work list, one ticket per group of G lanes (a group that finds the list exhausted idles while the
            // other one works; the wave leaves when every group has)
            // (the first entry of a group is its own index, as in sparse_add_kernel: tickets to one counter are served at ~23 ns each
            // device-wide, and at launch every group asks at once)
            int idx = base + r;
            if (base != (int)blockIdx.x * R) {
                if (i == 0) idx = (int)gridDim.x * R + atomicAdd(A.list_n + A.ticket_slot, 1);
                idx = __shfl(idx, r * G, 64);
            }
            valid = idx < A.list_n[0];
            patch = valid ? A.list[idx] : 0;
            if (!__builtin_amdgcn_ballot_w64(valid)) break;
        } else {
            if (base >= A.P) break;
            patch = base + r;
            valid = patch < A.P;
        }
        const int pc = valid ? patch : A.P - 1;
        const int o = A.off[pc], n = A.off[pc + 1] - o;
        int b = A.b[pc];
        int st = A.stat[pc];
        double* const alphag = A.alpha + (size_t)pc * ny * ldg;
        double* const BVg = A.BV + (size_t)pc * ldg * 2;
        double* const Cg = A.C + (size_t)pc * ldg * ldg;
        double* const Qg = A.Q + (size_t)pc * ldg * ldg;
        const unsigned long long rowmask = ((G == 64) ? ~0ull : ((1ull << G) - 1ull)) << (r * G);   // the lanes of this patch
        const int it0 = A.start_it ? A.start_it[pc] : 0;          // points of this call an earlier phase already took
        bool take = valid && n > 0 && it0 < n && b <= B;
        // state of the patch: rows in registers, blocks in LDS
        double al[3] = {0.0, 0.0, 0.0}, bv0 = 0.0, bv1 = 0.0;
        int ms = (i < b) ? sp_slot<QN>(i, b) : 0;                 // the slot of this lane's own column
        int p00 = b > 0 ? B * sp_slot<QN>(0, b) : 0;              // where C(0, 0) sits
        if (take && i < B) {
            // every slot starts from +0.0 -- rows, columns and vectors; the columns the basis has go to their slots
#pragma unroll
            for (int p = 0; p < B; ++p) {
                Cl[i + B * p] = 0.0;
                Ql[i + B * p] = 0.0;
            }
            kvL[i] = 0.0;
            svL[i] = 0.0;
            ehL[i] = 0.0;
            knL[i] = 0.0;
        }
        if (take && i < b) {
            bv0 = BVg[2 * i];
            bv1 = BVg[2 * i + 1];
            SP_FOR_C(c) al[c] = alphag[c * ldg + i];
            for (int j = 0; j < b; ++j) {
                const int pj = sp_slot<QN>(j, b);
                Cl[i + B * pj] = Cg[i + (size_t)j * ldg];
                Ql[i + B * pj] = Qg[i + (size_t)j * ldg];
            }
        }
        __builtin_amdgcn_wave_barrier();
        double Cr[REG ? B : 1], Qr[REG ? B : 1];
        double c00r = 0.0;                         // REG: C(0, 0) as lane 0 carries it (the NaN check of :245 reads it every point)
#pragma unroll
        for (int p = 0; p < (REG ? B : 1); ++p) { Cr[p] = 0.0; Qr[p] = 0.0; }
        auto rows_to_regs = [&]() {
            if constexpr (REG) {
#pragma unroll
                for (int p = 0; p < B; ++p) {
                    Cr[p] = Cl[i + B * p];
                    Qr[p] = Ql[i + B * p];
                }
                c00r = Cl[p00];
            }
        };
        auto rows_to_lds = [&]() {
            if constexpr (REG) {
#pragma unroll
                for (int p = 0; p < B; ++p) Cl[i + B * p] = Cr[p];
            }
        };
        if (take) rows_to_regs();
        {   // a state that already asks for a geometric deletion (possible only for one loaded with gpc_sparse_set_state) is not ours
            const bool asks = take && i < b && b > 1 && (double)1.0f / Ql[i + B * ms] < (double)1e-9f;
            if (__builtin_amdgcn_ballot_w64(asks) & rowmask) take = false;
        }
        if (valid && !take && i == 0) {
            if (LIST) {
                A.done_it[patch] = it0;                             // all of it is the later phases' work
                if (A.out_list && n > 0 && it0 < n) A.out_list[atomicAdd(A.out_list_n, 1)] = patch;
            } else {
                if (!A.start_it) A.done_it[patch] = 0;              // (a later phase leaves the earlier phase's count)
                if (A.list && n > 0 && it0 < n) A.list[atomicAdd(A.list_n, 1)] = patch;   // all of it is the later phases' work
                else if (A.list && A.status_out) A.status_out[patch] = st;                // nothing to do: nobody else visits it
            }
        }
        bool active = take;
        int it = it0, it_end = n;
        bool have_next = false;
        double kn_i = 0.0, pcn[4] = {0.0, 0.0, 0.0, 0.0}, pqn[4] = {0.0, 0.0, 0.0, 0.0};
        // the point in hand, and the next one in flight (two dependent loads: insertion order, then the point)
        double cx0 = 0.0, cx1 = 0.0, cy[3] = {0.0, 0.0, 0.0};
        int r_nxt = 0;
        if (active) {
            const int r0 = A.perm ? A.perm[o + it0] : it0;
            cx0 = A.x0[o + r0];
            cx1 = A.x1[o + r0];
            SP_FOR_C(c) cy[c] = A.y[(size_t)c * A.n_total + o + r0];
            if (it0 + 1 < n) r_nxt = A.perm ? A.perm[o + it0 + 1] : it0 + 1;
        }
        double nx0 = 0.0, nx1 = 0.0, nyv[3] = {0.0, 0.0, 0.0};
        int r_nxt2 = 0;
        // (bottom-tested: with the test at the top the compiler keeps a second copy of every loop-carried value for the exit path and
        // moves ~18 registers there and back per point)
        if (__builtin_amdgcn_ballot_w64(active)) do {
            if (active) {
                const double px0 = cx0, px1 = cx1;
                const double yv[3] = {cy[0], cy[1], cy[2]};
                const bool more = it + 1 < n;
                if (more) {
                    nx0 = A.x0[o + r_nxt];
                    nx1 = A.x1[o + r_nxt];
                    SP_FOR_C(c) nyv[c] = A.y[(size_t)c * A.n_total + o + r_nxt];
                    if (it + 2 < n) r_nxt2 = A.perm ? A.perm[o + it + 2] : it + 2;
                }
                const bool from_prev = have_next;
                have_next = false;
                bool stop = false;                  // hand the patch over before this point
                int dec = 0;
                if (b == 0) {
                    // First point (src/sparse_gp.hpp:100-114)
                    ms = sp_slot<QN>(i, 1);                                  // (meaningful for lane 0)
                    p00 = B * sp_slot<QN>(0, 1);
                    if (i == 0) {
                        SP_FOR_C(c) al[c] = yv[c] / (kstar + s20);
                        Cl[B * ms] = (double)(-1.0f) / (kstar + s20);
                        Ql[B * ms] = (double)(1.0f) / kstar;
                        bv0 = px0;
                        bv1 = px1;
                    }
                    __builtin_amdgcn_wave_barrier();
                    rows_to_regs();
                    b = 1;
                    dec = 0x81;
                } else {
                    // k, C k, e_hat = Q k (:119, :140, :160): from the previous point's update pass, or from scratch
                    // (one set of registers for both: the from-scratch form leaves its k and partial sums where the previous point's pass
                    // leaves them -- eight doubles and k copied per point otherwise)
                    if (!from_prev) {
                        kn_i = 0.0;
                        if (i < b) {
                            kn_i = gpc_rbf_neg(sf, A.c_exp, px0, px1, bv0, bv1, T);
                            kvL[ms] = kn_i;
                        }
                        __builtin_amdgcn_wave_barrier();
                        // the four quarters side by side, one slot of each per trip: their loads are issued together -- QN LDS round trips
                        // per pass instead of one per column; empty slots hold +0.0 (see sp_slot)
                        double acc_[4] = {0.0, 0.0, 0.0, 0.0}, acq_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll UNR
                        for (int t = 0; t < QN; ++t) {
                            double cv[4], qw[4], kj[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                kj[q] = kvL[QN * q + t];
                                cv[q] = REG ? Cr[REG ? QN * q + t : 0] : Cl[i + B * (QN * q + t)];
                                qw[q] = REG ? Qr[REG ? QN * q + t : 0] : Ql[i + B * (QN * q + t)];
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc_[q] += cv[q] * kj[q];
                                acq_[q] += qw[q] * kj[q];
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) { pcn[q] = acc_[q]; pqn[q] = acq_[q]; }
                    }
                    const double k_i = kn_i;
                    double ck_i = 0.0, eh_i = 0.0;
                    double dots[2] = {0.0, 0.0}, sums[3] = {0.0, 0.0, 0.0};
                    if (i < b) {
                        const double c_ = pcn[0] + pcn[1] + pcn[2] + pcn[3];
                        const double q_ = pqn[0] + pqn[1] + pqn[2] + pqn[3];
                        ck_i = c_;
                        eh_i = q_;
                        dots[0] += k_i * c_;                        // k^T C k   (:122)
                        dots[1] += k_i * q_;                        // k^T e_hat (:144)
                        SP_FOR_C(c) sums[c] += al[c] * k_i;   // m = alpha^T k (:121)
                    }
                    dots[0] = sp_row_sum<G>(dots[0]);
                    dots[1] = sp_row_sum<G>(dots[1]);
                    sums[0] = sp_row_sum<G>(sums[0]);
                    if (ny == 3) {
                        sums[1] = sp_row_sum<G>(sums[1]);
                        sums[2] = sp_row_sum<G>(sums[2]);
                    }
                    const double s2 = kstar + dots[0];
                    double gamma = kstar - dots[1];
                    if (gamma < (double)1e-12f) gamma = 0;          // :146-151
                    // gaussian_noise / gaussian_noise_3d (src/gaussian_noise.cpp:9-18, src/gaussian_noise_3d.cpp:11-20)
                    // (the divisions by the same denominator stay the compiler's: sharing the scaled denominator and its refined reciprocal
                    // between them -- 7 of the 13 operations of a division -- was worth 5 % of the colour GP's pass and agreed bit for bit on
                    // every fixed test, but tools/r4_stress_sparse.py found 2 of 300 random configurations, both ill-conditioned with
                    // eps_tol = 1e-14, whose branch decisions then differ from the other kernel shapes': not kept)
                    const double rr = (double)(-1.0f) / (s20 + s2);
                    double qv[3];
                    SP_FOR_C(c) qv[c] = (yv[c] - sums[c]) / (s20 + s2);
                    const bool fuse = A.fuse_next && more;
                    if (gamma < eps_tol && capacity != -1) {
                        // sparse update (:155-163)
                        const double eta = 1 / (1 + gamma * rr);
                        double sh = 0.0;
                        if (i < b) {
                            sh = ck_i + eh_i;                        // s_hat = C*k + e_hat
                            svL[ms] = sh;
                            SP_FOR_C(c) al[c] += sh * (qv[c] * eta);
                        }
                        const double re = rr * eta;
                        if constexpr (REG) c00r = c00r + (re * sh) * sh;          // lane 0: C(0, 0) exactly as the pass below forms it
                        if (fuse && i < b) {
                            kn_i = gpc_rbf_neg(sf, A.c_exp, nx0, nx1, bv0, bv1, T);
                            knL[ms] = kn_i;
                        }
                        __builtin_amdgcn_wave_barrier();
                        {
                            double acc_[4] = {0.0, 0.0, 0.0, 0.0}, acq_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll UNR
                            for (int t = 0; t < QN; ++t) if (B == G || i < B) {        // (lanes beyond the rows of state have no row)
                                // (no mask: an empty slot is 0 + x * 0 = +0.0 stored back and (+0.0)(+0.0) added to the sums, a row beyond
                                // the basis has s_hat = 0 and stays +0.0; without a next point the sums are never read)
                                double cv[4], qw[4], kj[4], sj[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    sj[q] = svL[QN * q + t];
                                    cv[q] = REG ? Cr[REG ? QN * q + t : 0] : Cl[i + B * (QN * q + t)];
                                    kj[q] = knL[QN * q + t];
                                    qw[q] = REG ? Qr[REG ? QN * q + t : 0] : Ql[i + B * (QN * q + t)];
                                }
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const double c = cv[q] + (re * sh) * sj[q];
                                    if constexpr (REG) Cr[REG ? QN * q + t : 0] = c;
                                    else Cl[i + B * (QN * q + t)] = c;
                                    acc_[q] += c * kj[q];
                                    acq_[q] += qw[q] * kj[q];
                                }
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) { pcn[q] = acc_[q]; pqn[q] = acq_[q]; }
                        }
                        have_next = fuse;
                    } else {
                        // full update (:164-203) -- if the basis stays within the block and no deletion follows it
                        const int nb = b + 1;
                        const double ig = (double)1.0f / gamma;
                        const double eh_x = (i < b) ? eh_i : (double)(-1.0f);
                        bool geo = false;
                        if (i < nb) {
                            const double q0 = (i < b) ? Ql[i + B * ms] : 0.0;
                            const double qd = q0 + (ig * eh_x) * eh_x;           // the updated diagonal of Q, as the update forms it
                            geo = (double)1.0f / qd < (double)1e-9f;              // :226-242 would delete
                        }
                        const bool any_geo = (__builtin_amdgcn_ballot_w64(geo) & rowmask) != 0;
                        if (nb > B || nb > ldg || (capacity > 0 && nb > capacity) || any_geo) {
                            stop = true;                    // nothing of this point has been applied
                        } else {
                            dec = 1;
                            rows_to_lds();                                       // (REG: the update and the move between slots happen in LDS)
                            const double si = (i < b) ? ck_i : (double)1.0f;
                            const int msn = sp_slot<QN>(i, nb);                  // this lane's slot in the basis of nb
                            if (i < nb) {
                                svL[msn] = si;
                                ehL[msn] = eh_x;
                                SP_FOR_C(c) {
                                    const double a0 = (i < b) ? al[c] : 0.0;
                                    al[c] = a0 + qv[c] * si;
                                }
                                if (i == b) { bv0 = px0; bv1 = px1; }
                                if (fuse) {
                                    kn_i = gpc_rbf_neg(sf, A.c_exp, nx0, nx1, bv0, bv1, T);
                                    knL[msn] = kn_i;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            {
                                // The update and the move to the slots of nb columns in one pass, quarter by quarter in COLUMN order: a
                                // column's new slot is never to the right of its old one and never the old slot of a later column, so a
                                // row is rewritten in place (the loads of a quarter precede its stores).  The sums come out in the same
                                // order as everywhere else: quarter q over its columns, ascending.
                                double acc_[4] = {0.0, 0.0, 0.0, 0.0}, acq_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int s0 = (nb * q) >> 2, s1 = (nb * (q + 1)) >> 2;
                                    double cv[QN], qw[QN];
                                    bool ok[QN];
#pragma unroll
                                    for (int t = 0; t < QN; ++t) {
                                        const int j = s0 + t;
                                        ok[t] = j < s1 && i < nb;
                                        const bool old = ok[t] && i < b && j < b;     // the new row / column starts from zero
                                        const int po = old ? sp_slot<QN>(j, b) : 0;
                                        cv[t] = old ? Cl[i + B * po] : 0.0;
                                        qw[t] = old ? Ql[i + B * po] : 0.0;
                                    }
#pragma unroll
                                    for (int t = 0; t < QN; ++t) {
                                        const int pn = QN * q + t;
                                        const double c = cv[t] + (rr * si) * svL[pn];              // (an empty slot: s = e_hat = k = +0.0)
                                        const double qn_ = qw[t] + (ig * eh_x) * ehL[pn];
                                        if (ok[t]) {
                                            Cl[i + B * pn] = c;
                                            Ql[i + B * pn] = qn_;
                                        }
                                        const double kj = knL[pn];
                                        acc_[q] += c * kj;
                                        acq_[q] += qn_ * kj;
                                    }
                                }
#pragma unroll
                                for (int q = 0; q < 4; ++q) { pcn[q] = acc_[q]; pqn[q] = acq_[q]; }
                            }
                            have_next = fuse;
                            b = nb;
                            ms = msn;
                            p00 = B * sp_slot<QN>(0, nb);
                            __builtin_amdgcn_wave_barrier();
                            rows_to_regs();
                        }
                    }
                }
                if (stop) {
                    it_end = it;
                    active = false;
                } else {
                    __builtin_amdgcn_wave_barrier();
                    // isnan(C(0,0)) -> "sparse_gp::C has become Nan" (:245)
                    const double c00 = REG ? c00r : Cl[p00];      // (REG: meaningful in lane 0, which is the one that writes the status)
                    if (c00 != c00 && st == GPC_STATUS_OK) st = GPC_STATUS_NAN;
                    if (A.trace && i == 0) A.trace[o + it] = (uint8_t)dec;
                    ++it;
                    cx0 = nx0; cx1 = nx1; cy[0] = nyv[0]; cy[1] = nyv[1]; cy[2] = nyv[2];
                    r_nxt = r_nxt2;
                    if (it >= n) active = false;
                }
            }
        } while (__builtin_amdgcn_ballot_w64(active));
        // write the state back (a patch handed over continues from it in the next kernel)
        if (take) {
            rows_to_lds();
            __builtin_amdgcn_wave_barrier();
            if (i < b) {
                BVg[2 * i] = bv0;
                BVg[2 * i + 1] = bv1;
                SP_FOR_C(c) alphag[c * ldg + i] = al[c];
                for (int j = 0; j < b; ++j) {
                    const int pj = sp_slot<QN>(j, b);
                    Cg[i + (size_t)j * ldg] = Cl[i + B * pj];
                    Qg[i + (size_t)j * ldg] = Ql[i + B * pj];
                }
            }
            if (i == 0) {
                if (!LIST && A.list && it_end < n) A.list[atomicAdd(A.list_n, 1)] = patch;    // handed over: the next phase continues it
                if (LIST && A.out_list && it_end < n) A.out_list[atomicAdd(A.out_list_n, 1)] = patch;
                A.done_it[patch] = it_end;
                A.b[patch] = b;
                A.count[patch] += it_end - it0;
                A.stat[patch] = st;
                if (A.status_out) A.status_out[patch] = st;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct SpPredParams {
    gpc_params prm;
    double c_exp;
    int P, ny, ld, m, conf;
    int fast;   // LDS holds a second [ld][SP_PC] buffer: V = C K by sp_ck_chunk
    const double *xs0, *xs1;
    const double *alpha, *C, *BV;
    const int32_t* b;
    double *f_star, *sigma;
    int32_t* status_out;
    const int32_t* stat;
    // ragged form (gpc_sparse_predict_points): patch i predicts at ITS OWN points off[i] .. off[i+1]-1 of xs0/xs1 and writes rows
    // off[i] .. of the output planes (plane stride n_total) -- predict_measurements(f, X_i, sigma) as the reference's training-set
    // RMS block calls it (/root/reference/src/gp_compressor.cpp:303-315).  nullptr: the shared grid of load_compressed.
    const int32_t* off;
    int n_total;
    int small_max;   // patches with at most this many basis vectors are the business of sparse_predict_small_kernel (-1: none)
};

#define SP_PC 32   // grid points per chunk of the sigma path

// V = C K for a chunk of SP_PC = 32 points on the MFMA pipe: C (b x b, global, column-major) times K (b x 32, LDS).
// v_mfma_f64_16x16x4_f64 with M = 16 rows of C, N = 16 points, K = 4 columns of C per instruction: the A operand of lane l
// is C[i0 + (l & 15)][j0 + (l >> 4)] (one 8-byte global load per lane, 16 contiguous rows per column), the B operand is
// K[j0 + (l >> 4)][p0 + (l & 15)] (one conflict-free LDS read).  Wave w owns the row tiles w, w+4, w+8, w+12 for both point
// tiles (8 accumulators); the loads of the next K-step are issued before the MFMAs of the current one, unconditionally
// (clamped addresses, masked values).  The result goes to LDS as Vc[row][point].  1300 MFMAs per chunk at b = 200.
// (The first version had every (point, column-group) thread walk its own columns of C with one broadcast global load and one
// LDS read per FMA: 1 TFLOP/s; a register-tiled VALU version was LDS-latency-bound with one wave per SIMD: 2.5 TFLOP/s.)
typedef double sp_d4 __attribute__((ext_vector_type(4)));
#define SP_RT 4   // row tiles per wave (4 waves x 4 x 16 rows = 256 = GPC_MAX_BV)
__device__ static inline void sp_ck_chunk(const double* __restrict__ Cg, int ld, int b, const double* Kc, double* Vc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int nrt = (b + 15) >> 4;
    sp_d4 acc[SP_RT][2];
#pragma unroll
    for (int t = 0; t < SP_RT; ++t) acc[t][0] = acc[t][1] = sp_d4{0.0, 0.0, 0.0, 0.0};
    int rowc[SP_RT];       // clamped row of this lane in tile t
    bool rowok[SP_RT];
#pragma unroll
    for (int t = 0; t < SP_RT; ++t) {
        const int i = 16 * (wave + 4 * t) + lr;
        rowok[t] = i < b;
        rowc[t] = min(i, b - 1);
    }
    // Round 4: the A operands of SP_PF K-steps are in flight (a K-step is 8 MFMAs = 512 cycles of the pipe per wave; with one step of
    // look-ahead every step waited out most of a ~2000-cycle load: the sigma path of a 200-vector basis ran at 0.18 of the FP64 peak)
    constexpr int SP_PF = 4;
    double an[SP_PF][SP_RT];
#pragma unroll
    for (int u = 0; u < SP_PF; ++u) {
        const int jc = min(4 * u + lg, b - 1);
#pragma unroll
        for (int t = 0; t < SP_RT; ++t) an[u][t] = Cg[rowc[t] + (size_t)jc * ld];
    }
    for (int jb = 0; jb < b; jb += 4 * SP_PF) {
#pragma unroll
        for (int u = 0; u < SP_PF; ++u) {
            const int j0 = jb + 4 * u;
            if (j0 < b) {                                  // (wave-uniform)
                const bool jok = j0 + lg < b;
                double ac[SP_RT];
#pragma unroll
                for (int t = 0; t < SP_RT; ++t) ac[t] = (jok && rowok[t]) ? an[u][t] : 0.0;
                {
                    const int jn = min(j0 + 4 * SP_PF + lg, b - 1);
#pragma unroll
                    for (int t = 0; t < SP_RT; ++t) an[u][t] = Cg[rowc[t] + (size_t)jn * ld];
                }
                const int jl = min(j0 + lg, b - 1);
                const double b0 = jok ? Kc[jl * SP_PC + lr] : 0.0;
                const double b1 = jok ? Kc[jl * SP_PC + 16 + lr] : 0.0;
#pragma unroll
                for (int t = 0; t < SP_RT; ++t) {
                    if (wave + 4 * t < nrt) {     // wave-uniform
                        acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], b0, acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], b1, acc[t][1], 0, 0, 0);
                    }
                }
            }
        }
    }
    // C/D layout: lane l, register r = V[i0 + (l >> 4) + 4 r][p0 + (l & 15)]
#pragma unroll
    for (int t = 0; t < SP_RT; ++t) {
        if (wave + 4 * t < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (wave + 4 * t) + lg + 4 * r;
                if (i < b) {
                    Vc[i * SP_PC + lr] = acc[t][0][r];
                    Vc[i * SP_PC + 16 + lr] = acc[t][1][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(SP_THREADS) void sparse_predict_kernel(SpPredParams A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int ld = A.ld, ny = A.ny;
    double* T = reinterpret_cast<double*>(smem);   // 64
    double* bv = T + 64;                           // 2*ld
    double* al = bv + 2 * ld;                      // ny*ld
    int* clamp = reinterpret_cast<int*>(al + 3 * ld);   // 2 doubles of room
    double* racc = al + 3 * ld + 2;                // [8][SP_PC]
    double* Kc = racc + 8 * SP_PC;                 // [ld][SP_PC]   (sigma path only; LDS is sized for it only then)
    double* Vc = Kc + (size_t)ld * SP_PC;          // [ld][SP_PC]   (A.fast only)
    gpc_exp_table_init(T);
    const double sf = A.prm.sigmaf_sq, s20 = A.prm.noise;

    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int b = A.b[patch];
        if (b <= A.small_max) continue;                 // (workgroup-uniform, before any barrier) sparse_predict_small_kernel took it
        const double* Cg = A.C + (size_t)patch * ld * ld;
        __syncthreads();
        for (int i = tid; i < b; i += SP_THREADS) {
            bv[2 * i] = A.BV[(size_t)patch * ld * 2 + 2 * i];
            bv[2 * i + 1] = A.BV[(size_t)patch * ld * 2 + 2 * i + 1];
            for (int c = 0; c < ny; ++c) al[c * ld + i] = A.alpha[((size_t)patch * ny + c) * ld + i];
        }
        if (tid == 0) *clamp = 0;
        __syncthreads();
        const int po = A.off ? A.off[patch] : 0;                           // first point of this patch in xs0 / xs1
        const int m = A.off ? A.off[patch + 1] - po : A.m;
        const size_t fstride = A.off ? (size_t)A.n_total : (size_t)m;     // distance between the output planes
        const double* xs0 = A.xs0 + po;
        const double* xs1 = A.xs1 + po;
        double* fs = A.off ? A.f_star + po : A.f_star + (size_t)patch * ny * m;
        // mean: f = alpha^T k (:329); b == 0 -> 0 (:321-327)
        for (int p = tid; p < m; p += SP_THREADS) {
            const double q0 = xs0[p], q1 = xs1[p];
            double s[3] = {0.0, 0.0, 0.0};
            for (int i = 0; i < b; ++i) {
                const double k = gpc_rbf_neg(sf, A.c_exp, q0, q1, bv[2 * i], bv[2 * i + 1], T);
                for (int c = 0; c < ny; ++c) s[c] += al[c * ld + i] * k;
            }
            for (int c = 0; c < ny; ++c) fs[(size_t)c * fstride + p] = s[c];
        }
        if (A.sigma) {
            double* sg = A.off ? A.sigma + po : A.sigma + (size_t)patch * m;
            const double kstar = sf;
            for (int p0 = 0; p0 < m; p0 += SP_PC) {
                const int pc = min(SP_PC, m - p0);
                __syncthreads();
                for (int e = tid; e < b * SP_PC; e += SP_THREADS) {
                    const int pp = e & (SP_PC - 1), i = e / SP_PC;
                    Kc[i * SP_PC + pp] = (pp < pc) ? gpc_rbf_neg(sf, A.c_exp, xs0[p0 + pp], xs1[p0 + pp], bv[2 * i], bv[2 * i + 1], T) : 0.0;
                }
                __syncthreads();
                if (A.fast) {
                    sp_ck_chunk(Cg, ld, b, Kc, Vc);
                    __syncthreads();
                }
                const int pp = tid & (SP_PC - 1), ig = tid / SP_PC;   // 8 row groups
                double acc = 0.0;
                for (int j = ig; j < b; j += SP_THREADS / SP_PC) {
                    // (C k)_j  (:330; C is symmetric)
                    double t = 0.0;
                    if (A.fast) t = Vc[j * SP_PC + pp];
                    else
                        for (int i = 0; i < b; ++i) t += Kc[i * SP_PC + pp] * Cg[i + (size_t)j * ld];
                    acc += t * Kc[j * SP_PC + pp];
                }
                racc[ig * SP_PC + pp] = acc;
                __syncthreads();
                if (tid < pc) {
                    double kCk = 0.0;
                    for (int q = 0; q < SP_THREADS / SP_PC; ++q) kCk += racc[q * SP_PC + tid];
                    double sigma = (b == 0) ? kstar + s20 : s20 + kstar + kCk;
                    if (sigma < 0) { sigma = 0; *clamp = 1; }                 // :334-337
                    if (A.conf) {
                        sigma /= kstar + s20;
                        sigma = (double)100.0f * ((double)1.0f - sigma);    // :340-345
                    } else {
                        sigma = sqrt(sigma);
                    }
                    sg[p0 + tid] = sigma;
                }
            }
        }
        __syncthreads();
        if (tid == 0 && A.status_out) {
            int st = A.stat[patch];
            if (st == GPC_STATUS_OK && *clamp) st = GPC_STATUS_SIGMA_CLAMPED;
            A.status_out[patch] = st;
        }
    }
}

// ---- predict with a SMALL basis: one wave per patch, a lane per grid point (round 4) ------------------------------------------------
// At the reference's default hyper-parameters a patch keeps ~13 basis vectors (8 .. 41 over a batch), and predict_measurements ALWAYS
// computes sigma = sqrt(s20 + k* + k^T C k) (/root/reference/src/sparse_gp.hpp:299-351; the caller drops it, src/gp_compressor.cpp:333-334).
// sparse_predict_kernel is shaped for a basis of 100 .. 200 -- a 256-thread workgroup per patch, chunks of 32 points, K and V = C K
// through LDS, the MFMA pipe, five barriers per chunk -- and at b = 13 its sigma path took 4.3 ms for 32768 patches (the mean 0.5 ms):
// 1.2 TFLOP/s on 6 GFLOP.  Here a lane owns a grid point: its b kernel values stay in registers (BM = 16 or 32 of them, zero beyond b),
// C sits in LDS zero-padded to BM x BM and is read by broadcast, the mean and k^T C k are register FMAs -- no barrier, no reduction, no
// second evaluation of k.  Mean: the same operations in the same order as sparse_predict_kernel (bit-identical); sigma: t_j = sum_i
// C_ij k_i, then sum_j t_j k_j, a summation order of its own, held by the tolerance against the oracle.  Patches with more than BM
// vectors are left to sparse_predict_kernel (SpPredParams::small_max), patches within the other instance's range to that one.
template <int BM>
__global__ __launch_bounds__(64) void sparse_predict_small_kernel(SpPredParams A, int b_lo)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* T = reinterpret_cast<double*>(smem);   // 64
    double* Cl = T + 64;                           // [BM][BM] column-major, zero-padded
    double* al = Cl + BM * BM;                     // [3][BM]
    double* bv = al + 3 * BM;                      // [BM][2]
    const int lane = threadIdx.x;
    const int ld = A.ld, ny = A.ny;
    gpc_exp_table_init(T);
    const double sf = A.prm.sigmaf_sq, s20 = A.prm.noise, kstar = sf;
    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(A.b[patch]);
        if (b < b_lo || b > BM) continue;
        __builtin_amdgcn_wave_barrier();           // (one wave: LDS instructions execute in order; the compiler must keep them so)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const double* Cg = A.C + (size_t)patch * ld * ld;
        for (int e = lane; e < BM * BM; e += 64) {
            const int i = e % BM, j = e / BM;
            Cl[e] = (i < b && j < b) ? Cg[i + (size_t)j * ld] : 0.0;
        }
        if (lane < BM) {
            const bool in = lane < b;
            bv[2 * lane] = in ? A.BV[(size_t)patch * ld * 2 + 2 * lane] : 0.0;
            bv[2 * lane + 1] = in ? A.BV[(size_t)patch * ld * 2 + 2 * lane + 1] : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) al[c * BM + lane] = (in && c < ny) ? A.alpha[((size_t)patch * ny + c) * ld + lane] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int po = A.off ? A.off[patch] : 0;
        const int m = A.off ? A.off[patch + 1] - po : A.m;
        const size_t fstride = A.off ? (size_t)A.n_total : (size_t)m;
        const double* xs0 = A.xs0 + po;
        const double* xs1 = A.xs1 + po;
        double* fs = A.off ? A.f_star + po : A.f_star + (size_t)patch * ny * m;
        double* sg = A.sigma ? (A.off ? A.sigma + po : A.sigma + (size_t)patch * m) : nullptr;
        bool clamped = false;
        // (the lane's grid coordinates are loaded per iteration, on purpose: holding a shared grid in registers -- 28 VGPRs, two waves per
        // SIMD less -- measured 1.24 against 1.12 ms for the sigma-predict of the defaults batch, staging it in LDS once per wave 1.30)
        double nq0 = 0.0, nq1 = 0.0;             // the NEXT 64 points' coordinates are requested before this block's arithmetic
        if (lane < m) { nq0 = xs0[lane]; nq1 = xs1[lane]; }
        for (int p = lane; p < m; p += 64) {
            const double q0 = nq0, q1 = nq1;
            if (p + 64 < m) { nq0 = xs0[p + 64]; nq1 = xs1[p + 64]; }
            double k[BM];
            double s[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < BM; ++i) {
                k[i] = 0.0;
                if (i < b) {                                    // (wave-uniform)
                    k[i] = gpc_rbf_neg(sf, A.c_exp, q0, q1, bv[2 * i], bv[2 * i + 1], T);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ny) s[c] += al[c * BM + i] * k[i];   // f = alpha^T k (:329), in sparse_predict_kernel's order
                }
            }
            for (int c = 0; c < ny; ++c) fs[(size_t)c * fstride + p] = s[c];
            if (sg) {
                double kCk = 0.0;
#pragma unroll
                for (int j = 0; j < BM; ++j) {
                    if (j < b) {                                // (wave-uniform)
                        double t = 0.0;
#pragma unroll
                        for (int i = 0; i < BM; ++i) t += k[i] * Cl[i + BM * j];     // (C k)_j (:330; rows beyond b are zero)
                        kCk += t * k[j];
                    }
                }
                double sigma = (b == 0) ? kstar + s20 : s20 + kstar + kCk;
                if (sigma < 0) { sigma = 0; clamped = true; }                 // :334-337
                if (A.conf) {
                    sigma /= kstar + s20;
                    sigma = (double)100.0f * ((double)1.0f - sigma);    // :340-345
                } else {
                    sigma = sqrt(sigma);
                }
                sg[p] = sigma;
            }
        }
        const bool any_clamp = __builtin_amdgcn_ballot_w64(clamped) != 0;
        if (lane == 0 && A.status_out) {
            int st = A.stat[patch];
            if (st == GPC_STATUS_OK && any_clamp) st = GPC_STATUS_SIGMA_CLAMPED;
            A.status_out[patch] = st;
        }
    }
}

// ---- registration inner loop: likelihoods and their derivatives on ragged point sets ----
// sparse_gp::compute_likelihoods -> likelihood (/root/reference/src/sparse_gp.hpp:387-427) and compute_derivatives ->
// likelihood_dx (:463-508) with rbf_kernel::kernel_dx (src/rbf_kernel.cpp:33-41); field variants
// src/sparse_gp_field.hpp:322-392.  Per point: k (b), v = C k (the O(b^2) part), then
//   sigma = s20 + k^T v + k**,  off = y - alpha^T k,  sigma_dx = 2 k_dx^T v,  k_dx row j = -(p0/p1) (x - BV_j) exp(..) = -(x - BV_j) k_j / p1
//   l = exp(-|off|^2 / (2 sigma)) / sqrt((2 pi)^ny sigma)
//   dX = exppart * (-sigma_dx + 2 (k_dx^T alpha) off + sigma_dx / sigma |off|^2),  exppart = exp(-|off|^2/(2 sigma)) / (2 sigma^1.5)
// Same work distribution as the sigma path of sparse_predict_kernel: chunks of SP_PC points, thread = (point, one of 8
// row groups of C), partial sums reduced through LDS.
struct SpLikParams {
    gpc_params prm;
    double c_exp;
    int P, ny, ld, n_total;
    int fast;   // LDS holds a second [ld][SP_PC] buffer: V = C K by sp_ck_chunk
    const int32_t* off;
    const double *x0, *x1, *y;
    const double *alpha, *C, *BV;
    const int32_t* b;
    double *dX, *l;
    double* raw;   // train_sigmaf pass (prm.sigmaf_sq == 1): per point e^T C e, alpha^T e, sum_j |x - BV_j|^2 e_j alpha_j
};
#define SP_NQ 12   // partial sums per thread: kCk, 2 x (k_dx^T v), ny x mu, 2 x ny x (k_dx^T alpha)

__global__ __launch_bounds__(SP_THREADS) void sparse_likelihood_kernel(SpLikParams A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int ld = A.ld, ny = A.ny;
    double* T = reinterpret_cast<double*>(smem);   // 64
    double* bv = T + 64;                           // 2*ld
    double* al = bv + 2 * ld;                      // 3*ld
    double* racc = al + 3 * ld;                    // [SP_NQ][8][SP_PC]
    double* Kc = racc + SP_NQ * 8 * SP_PC;         // [ld][SP_PC]
    double* Vc = Kc + (size_t)ld * SP_PC;          // [ld][SP_PC]   (A.fast only)
    gpc_exp_table_init(T);
    const double sf = A.prm.sigmaf_sq, s20 = A.prm.noise, inv_l = 1.0 / A.prm.l_sq;

    for (int patch = blockIdx.x; patch < A.P; patch += gridDim.x) {
        const int b = A.b[patch];
        const int o = A.off[patch], n = A.off[patch + 1] - o;
        const double* Cg = A.C + (size_t)patch * ld * ld;
        __syncthreads();
        for (int i = tid; i < b; i += SP_THREADS) {
            bv[2 * i] = A.BV[(size_t)patch * ld * 2 + 2 * i];
            bv[2 * i + 1] = A.BV[(size_t)patch * ld * 2 + 2 * i + 1];
            for (int c = 0; c < ny; ++c) al[c * ld + i] = A.alpha[((size_t)patch * ny + c) * ld + i];
        }
        for (int p0 = 0; p0 < n; p0 += SP_PC) {
            const int pc = min(SP_PC, n - p0);
            __syncthreads();
            for (int e = tid; e < b * SP_PC; e += SP_THREADS) {
                const int pp = e & (SP_PC - 1), i = e / SP_PC;
                Kc[i * SP_PC + pp] = (pp < pc) ? gpc_rbf_neg(sf, A.c_exp, A.x0[o + p0 + pp], A.x1[o + p0 + pp], bv[2 * i], bv[2 * i + 1], T) : 0.0;
            }
            __syncthreads();
            if (A.fast) {
                sp_ck_chunk(Cg, ld, b, Kc, Vc);
                __syncthreads();
            }
            const int pp = tid & (SP_PC - 1), ig = tid / SP_PC;   // 8 row groups
            const bool live = pp < pc;
            const double q0 = live ? A.x0[o + p0 + pp] : 0.0, q1 = live ? A.x1[o + p0 + pp] : 0.0;
            double acc[SP_NQ];
#pragma unroll
            for (int q = 0; q < SP_NQ; ++q) acc[q] = 0.0;
            for (int j = ig; j < b; j += SP_THREADS / SP_PC) {
                double t = 0.0;                                   // v_j = (C k)_j, C symmetric
                if (A.fast) t = Vc[j * SP_PC + pp];
                else
                    for (int i = 0; i < b; ++i) t += Kc[i * SP_PC + pp] * Cg[i + (size_t)j * ld];
                const double kj = Kc[j * SP_PC + pp];
                const double g0 = -(q0 - bv[2 * j]) * kj * inv_l, g1 = -(q1 - bv[2 * j + 1]) * kj * inv_l;   // k_dx row j
                acc[0] += t * kj;
                acc[1] += g0 * t;
                acc[2] += g1 * t;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (c < ny) {
                        const double a = al[c * ld + j];
                        acc[3 + c] += a * kj;
                        acc[6 + c] += g0 * a;
                        acc[9 + c] += g1 * a;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < SP_NQ; ++q) racc[(q * 8 + ig) * SP_PC + pp] = acc[q];
            __syncthreads();
            if (tid < pc) {
                double r[SP_NQ];
#pragma unroll
                for (int q = 0; q < SP_NQ; ++q) {
                    double s_ = 0.0;
                    for (int w = 0; w < SP_THREADS / SP_PC; ++w) s_ += racc[(q * 8 + w) * SP_PC + tid];
                    r[q] = s_;
                }
                const double kstar = sf;
                double offv[3] = {0.0, 0.0, 0.0}, sq = 0.0;
                for (int c = 0; c < ny; ++c) {
                    offv[c] = A.y[(size_t)c * A.n_total + o + p0 + tid] - r[3 + c];
                    sq += offv[c] * offv[c];
                }
                if (A.l) {
                    const double sigma = s20 + kstar + r[0];                                    // :420-425
                    const double two_pi = (double)2.0f * 3.14159265358979323846;
                    const double norm = (ny == 1) ? two_pi * sigma : two_pi * two_pi * two_pi * sigma;
                    A.l[o + p0 + tid] = (double)1.0f / sqrt(norm) * exp((double)(-0.5f) / sigma * sq);
                }
                if (A.dX) {
                    const double sigma = s20 + r[0] + kstar;                                    // :485
                    const double sqrtsigma = sqrt(sigma);
                    const double exppart = (double)0.5f / (sigma * sqrtsigma) * exp((double)(-0.5f) / sigma * sq);
                    double* d = A.dX + (size_t)(o + p0 + tid) * 3;
                    for (int dd = 0; dd < 2; ++dd) {
                        const double sigma_dx = (double)2.0f * r[1 + dd];
                        double ko = 0.0;
                        for (int c = 0; c < ny; ++c) ko += r[6 + 3 * dd + c] * offv[c];
                        d[1 + dd] = exppart * (-sigma_dx + (double)2.0f * ko + sigma_dx / sigma * sq);
                    }
                    d[0] = (ny == 1) ? (double)(-1.0f) / (sigma * sqrtsigma) * offv[0] * exppart : 0.0;   // field: dx(0) = 0
                }
                if (A.raw) {
                    const double u0 = A.x0[o + p0 + tid], u1 = A.x1[o + p0 + tid];
                    double h = 0.0;
                    for (int j = 0; j < b; ++j) {
                        const double d0 = u0 - bv[2 * j], d1 = u1 - bv[2 * j + 1];
                        h += (d0 * d0 + d1 * d1) * Kc[j * SP_PC + tid] * al[j];
                    }
                    double* w = A.raw + (size_t)(o + p0 + tid) * 3;
                    w[0] = r[0]; w[1] = r[3]; w[2] = h;
                }
            }
        }
    }
}

// ---- row f4: the live part of sparse_gp::train_parameters (src/sparse_gp.hpp:586-640) ---------------------------------
// The inner do-loop holds the state (alpha, C, BV) fixed and moves only kernel.param()(0) = sigma_f^2 = p, and every
// quantity it evaluates is a polynomial in p over per-point sums that do not depend on p:
//     k = p e,   alpha^T k = p a_i,   k_dtheta(:,0)^T alpha = a_i,   k_dtheta(:,1)^T alpha = p 0.5f/p1^2 h_i,   k^T C k = p^2 q_i
// with e_j = exp(-0.5f/p1 |x_i - BV_j|^2), a_i = alpha^T e, h_i = sum_j |x_i - BV_j|^2 e_j alpha_j, q_i = e^T C e.  The O(n b^2)
// sums come from one pass of sparse_likelihood_kernel (MFMA C K) with sigma_f^2 = 1; the <= 102 iterations are then O(n)
// each and run here, one wave per patch.
struct SpTrainParams {
    int P, max_counter;
    double sf, l_sq, s20, step;
    const int32_t *off, *b;
    const double *raw, *y;
    double *p0, *ls, *delta;
    int32_t* iters;
};

__device__ static inline double sp_wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);    // the same bits in every lane
    return v;
}

__global__ __launch_bounds__(SP_THREADS) void sparse_train_kernel(SpTrainParams A)
{
    const int lane = threadIdx.x & 63;
    const int patch = blockIdx.x * (SP_THREADS / 64) + (threadIdx.x >> 6);
    if (patch >= A.P) return;
    const int o = A.off[patch], n = A.off[patch + 1] - o;
    double p = A.sf, d0 = 0.0, d1 = 0.0;
    int iters = 0;
    if (A.b[patch] >= 20) {                                        // "if (first && BV.cols() < 20) return;"  (:609-611)
        const double logsqrt2pi = (double)0.5f * log((double)2.0f * 3.14159265358979323846);
        const double hc = (double)0.5f / (A.l_sq * A.l_sq);
        const double* raw = A.raw + (size_t)o * 3;
        const double* y = A.y + o;
        int counter = 0;
        do {
            d0 = d1 = 0.0;
            for (int i = lane; i < n; i += 64) {                   // likelihood_dtheta (:510-519), summed over the points (:619-623)
                const double a = raw[3 * i + 1], h = raw[3 * i + 2];
                const double r = p * a - y[i];
                d0 += r * a;
                d1 += r * (p * hc * h);
            }
            d0 = sp_wave_sum(d0);
            d1 = sp_wave_sum(d1);
            p += A.step * d0;                                      // :624
            double ls = 0.0;
            for (int i = lane; i < n; i += 64) {                   // log_likelihood (:356-385) with the updated parameter
                const double q = raw[3 * i], a = raw[3 * i + 1];
                const double mu = p * a, sigma = A.s20 + p + p * p * q;
                const double cent2 = (y[i] - mu) * (y[i] - mu);
                ls += -logsqrt2pi - (double)0.5f * log(sigma) - (double)0.5f * cent2 / sigma;
            }
            ls = sp_wave_sum(ls);
            if (lane == 0) A.ls[(size_t)patch * (A.max_counter + 2) + counter] = ls;
            iters = counter + 1;
            if (counter > A.max_counter) break;                    // :630-633
            ++counter;
        } while (sqrt(d0 * d0 + d1 * d1) > (double)1e-2f);         // :636 (a NaN gradient ends the loop as well)
    }
    if (lane == 0) {
        A.p0[patch] = p;
        A.iters[patch] = iters;
        A.delta[2 * patch] = d0;
        A.delta[2 * patch + 1] = d1;
    }
}

// ------------------------------------------------------------------------------------------------ host side

static size_t sp_add_lds_small(int bm = SP_BMAX) { return sizeof(double) * (size_t)(64 + 16 + 6 + 4 * (bm + 2) + 22 * (bm + 1) + 2 * bm * bm); }
static size_t sp_add_lds(int ld, bool tri = false) { return sizeof(double) * (size_t)(64 + 16 + 6 + 4 * (ld + 1) + 8 * ld + 9 * ld + 5 * ld + (tri ? 4 * ld : 0)); }
static size_t sp_lik_lds(int ld, bool fast)
{
    return sizeof(double) * (size_t)(64 + 5 * ld + SP_NQ * 8 * SP_PC + (size_t)ld * SP_PC * (fast ? 2 : 1));
}
static size_t sp_pred_lds(int ld, bool sigma, bool fast)
{
    return sizeof(double) * (size_t)(64 + 5 * ld + (sigma ? (size_t)ld * SP_PC * (fast ? 2 : 1) : 0) + 8 * SP_PC + 2);
}

// The two instances of the small-basis predict kernel (b <= 16, 17 .. 32), ahead of sparse_predict_kernel on the same stream; sets
// A.small_max so that the regular kernel skips what they took.  GPC_SPARSE_NO_SMALL_PREDICT=1: everything through the regular kernel.
// The predict launches: the two instances of the small-basis kernel (b <= 16, 17 .. 32) and sparse_predict_kernel for the rest (it skips what
// they took: A.small_max).  The three work on disjoint patches, and at the reference's defaults each is a short launch that ends in a tail of a few
// long patches: they run SIDE BY SIDE on the context's stream and its two copy streams (forked and joined with events; the streams exist since
// gpc_ctx_create).  GPC_SPARSE_NO_SMALL_PREDICT=1: everything through the regular kernel; GPC_SPARSE_PREDICT_NO_FORK=1: one after the other.
static int sp_predict_launch(gpc_ctx* ctx, SpPredParams& A, int grid, size_t lds)
{
    A.small_max = -1;
    hipStream_t main_s = ctx->stream;
    if (getenv("GPC_SPARSE_NO_SMALL_PREDICT") || A.ld < 1) {
        hipLaunchKernelGGL(sparse_predict_kernel, dim3(grid), dim3(SP_THREADS), lds, main_s, A);
        GPC_HIP(ctx, hipGetLastError());
        return GPC_OK;
    }
    // (with sigma only: measured at the reference's defaults, 32768 patches -- sigma-predict 1.14 -> 0.84 ms; the mean-only launches are too
    // short to pay for the fork and join, 0.23 -> 0.28 ms)
    const bool fork = ctx->s_in && ctx->s_out && A.sigma != nullptr && !getenv("GPC_SPARSE_PREDICT_NO_FORK");
    hipStream_t s32 = fork ? ctx->s_in : main_s, sreg = fork ? ctx->s_out : main_s;
    // (the third launch goes to the context's OWN stream: the legacy default stream does not overlap its kernels with another stream's, and
    // a caller's stream may share a hardware queue with s_in or s_out -- gpc_api.hip, dense_host; own_stream, s_in and s_out never do)
    hipStream_t s16 = (fork && ctx->own_stream) ? ctx->own_stream : main_s;
    if (fork) {
        GPC_HIP(ctx, hipEventRecord(ctx->ev[0][14], main_s));
        GPC_HIP(ctx, hipStreamWaitEvent(s32, ctx->ev[0][14], 0));
        GPC_HIP(ctx, hipStreamWaitEvent(sreg, ctx->ev[0][14], 0));
        if (s16 != main_s) GPC_HIP(ctx, hipStreamWaitEvent(s16, ctx->ev[0][14], 0));
    }
    const int waves = std::min(A.P, ctx->num_cus * (getenv("GPC_SP_SMALL_WAVES") ? atoi(getenv("GPC_SP_SMALL_WAVES")) : 16));
    const size_t l16 = sizeof(double) * (size_t)(64 + 16 * 16 + 5 * 16), l32 = sizeof(double) * (size_t)(64 + 32 * 32 + 5 * 32);
    A.small_max = 32;
    // (the regular kernel first: its few patches are the longest)
    hipLaunchKernelGGL(sparse_predict_kernel, dim3(grid), dim3(SP_THREADS), lds, sreg, A);
    GPC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((sparse_predict_small_kernel<32>), dim3(std::min(A.P, ctx->num_cus * 12)), dim3(64), l32, s32, A, 17);
    GPC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((sparse_predict_small_kernel<16>), dim3(waves), dim3(64), l16, s16, A, 0);
    GPC_HIP(ctx, hipGetLastError());
    if (fork) {
        GPC_HIP(ctx, hipEventRecord(ctx->ev[1][14], s32));
        GPC_HIP(ctx, hipEventRecord(ctx->ev[2][14], sreg));
        GPC_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev[1][14], 0));
        GPC_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev[2][14], 0));
        if (s16 != main_s) {
            GPC_HIP(ctx, hipEventRecord(ctx->ev[1][12], s16));
            GPC_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev[1][12], 0));
        }
    }
    return GPC_OK;
}

extern "C" {

int gpc_sparse_create(gpc_ctx* ctx, const gpc_params* params, int P, int ny, gpc_sparse** out)
{
    if (!ctx || ctx->dead.load()) return GPC_EINVAL;
    if (!out) return gpc_fail(ctx, GPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (!params) return gpc_fail(ctx, GPC_EINVAL, "params is NULL");
    if (P < 0) return gpc_fail(ctx, GPC_EINVAL, "negative P");
    if (ny != 1 && ny != 3) return gpc_fail(ctx, GPC_EINVAL, "ny must be 1 (sparse_gp) or 3 (sparse_gp_field), got %d", ny);
    if (params->capacity == 0 || params->capacity < -1) return gpc_fail(ctx, GPC_EINVAL, "capacity must be > 0 or -1");
    if (params->capacity > GPC_MAX_BV - 1) return gpc_fail(ctx, GPC_ERANGE, "capacity %d > %d", params->capacity, GPC_MAX_BV - 1);
    if (params->noise_model < 0 || params->noise_model > 2) return gpc_fail(ctx, GPC_EINVAL, "noise_model must be 0, 1 or 2");
    if (params->noise_model != 0 && ny != 1) return gpc_fail(ctx, GPC_EINVAL, "probit noise needs ny == 1");
    if (!(params->l_sq > 0.0) || !(params->sigmaf_sq > 0.0)) return gpc_fail(ctx, GPC_EINVAL, "kernel parameters out of range");
    gpc_sparse* g = new (std::nothrow) gpc_sparse();
    if (!g) return GPC_ENOMEM;
    g->ctx = ctx; g->prm = *params; g->P = P; g->ny = ny;
    // capacity + 1 rows (a full update holds capacity + 1 basis vectors until the deletion that follows it), rounded up to 16
    // doubles = 128 B: every column of C and Q then starts on a cache line, and a wave's 64-row segment is exactly 4 lines.
    // With ld = 201 the segments straddled lines -- 5 fetched per 4 used, re-fetched from HBM by the next row trip -- and
    // the add path, which is bound by exactly this stream, read 27 % more than it consumed (profiles/r02_summary.json).
    g->ld = params->capacity == -1 ? GPC_MAX_BV : ((params->capacity + 1 + 15) & ~15);
    g->alpha = g->C = g->Q = g->BV = nullptr;
    g->b = g->count = g->stat = nullptr;
    g->done_it = nullptr;
    g->list = nullptr;
    g->trace = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t ld = (size_t)g->ld, Pn = (size_t)(P > 0 ? P : 1);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&g->alpha, 8 * Pn * ny * ld);
    if (e == hipSuccess) e = hipMalloc(&g->C, 8 * Pn * ld * ld);
    if (e == hipSuccess) e = hipMalloc(&g->Q, 8 * Pn * ld * ld);
    if (e == hipSuccess) e = hipMalloc(&g->BV, 8 * Pn * ld * 2);
    if (e == hipSuccess) e = hipMalloc(&g->b, 4 * Pn);
    if (e == hipSuccess) e = hipMalloc(&g->count, 4 * Pn);
    if (e == hipSuccess) e = hipMalloc(&g->stat, 4 * Pn);
    if (e == hipSuccess) e = hipMalloc(&g->done_it, 4 * Pn);
    if (e == hipSuccess) e = hipMalloc(&g->list, 4 * (3 * Pn + 12));      // three work lists of P entries, then their 3 x 4 counters
    if (e == hipSuccess) e = hipMemsetAsync(g->b, 0, 4 * Pn, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->count, 0, 4 * Pn, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->stat, 0, 4 * Pn, ctx->stream);
    if (e != hipSuccess) {
        int rc = gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_create: %s", hipGetErrorString(e));
        for (void* p : {(void*)g->alpha, (void*)g->C, (void*)g->Q, (void*)g->BV, (void*)g->b, (void*)g->count, (void*)g->stat, (void*)g->done_it, (void*)g->list})
            if (p) (void)hipFree(p);
        delete g;
        return rc;
    }
    gpc_ctx_ref(ctx);
    *out = g;
    return GPC_OK;
}

// Safe in either order with gpc_ctx_destroy: a context destroyed first has synchronised its stream already and stays
// allocated (dead) until its last child is gone.
void gpc_sparse_destroy(gpc_sparse* g)
{
    if (!g) return;
    gpc_ctx* ctx = g->ctx;
    (void)hipSetDevice(ctx->device);
    if (!ctx->dead.load()) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->dead.load()) (void)hipStreamSynchronize(ctx->stream);
    }
    for (void* p : {(void*)g->alpha, (void*)g->C, (void*)g->Q, (void*)g->BV, (void*)g->b, (void*)g->count, (void*)g->stat, (void*)g->done_it, (void*)g->list})
        if (p) (void)hipFree(p);
    delete g;
    gpc_ctx_unref(ctx);
}

int gpc_sparse_ld(const gpc_sparse* g) { return g ? g->ld : GPC_EINVAL; }

int gpc_sparse_set_trace(gpc_sparse* g, uint8_t* trace_dev)
{
    if (!g) return GPC_EINVAL;
    if (g->ctx->dead.load()) return GPC_EINVAL;
    std::lock_guard<std::mutex> lk(g->ctx->mu);
    g->trace = trace_dev;
    return GPC_OK;
}

int gpc_sparse_reset(gpc_sparse* g)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t Pn = (size_t)(g->P > 0 ? g->P : 1);
    GPC_HIP(ctx, hipMemsetAsync(g->b, 0, 4 * Pn, ctx->stream));
    GPC_HIP(ctx, hipMemsetAsync(g->count, 0, 4 * Pn, ctx->stream));
    GPC_HIP(ctx, hipMemsetAsync(g->stat, 0, 4 * Pn, ctx->stream));
    return GPC_OK;
}

int gpc_sparse_add_dev(gpc_sparse* g, const int32_t* off, int n_max, int n_total, const double* x0, const double* x1,
                       const double* y, const int32_t* perm, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (n_total < 0 || n_max < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (n_total > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (g->P == 0) return GPC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    SpAddParams A;
    A.prm = g->prm;
    A.c_exp = (double)(-0.5f) / g->prm.l_sq;
    A.P = g->P; A.ny = g->ny; A.ld = g->ld; A.n_total = n_total;
    A.off = off; A.x0 = x0; A.x1 = x1; A.y = y; A.perm = perm;
    A.alpha = g->alpha; A.C = g->C; A.Q = g->Q; A.BV = g->BV;
    A.b = g->b; A.count = g->count; A.stat = g->stat; A.status_out = status;
    A.fuse_next = getenv("GPC_SPARSE_NO_FUSE") ? 0 : 1;
    A.trace = g->trace;

    // capacity <= 64: one wave per patch (every row of the basis fits a lane; no cross-wave barriers, four times the patches
    // in flight); capacity <= 100 (the reference's default): two waves, twice the patches in flight; otherwise four waves per
    // patch.  While a thread owns at most one basis row the shapes give the same results, bit for bit (GPC_SPARSE_WIDE forces
    // the four-wave shape: tests).  Measured at capacity 100: 465 k vs 268 k patches/s in the small-basis regime, 75 k vs 68 k
    // with a full basis; at capacity 128 two waves lose (49 k vs 54 k).
    const int cap_ = g->prm.capacity;
    const int nth0 = (cap_ <= 0 || getenv("GPC_SPARSE_WIDE")) ? SP_THREADS : cap_ <= 64 ? 64 : cap_ <= 100 ? 128 : SP_THREADS;
    // four or two waves per patch (capacity > 64): the triangular mode (sp_tri_pass) -- half the stream of C and Q; GPC_SPARSE_FULL=1 keeps the full passes
    const bool tri = nth0 >= 128 && A.prm.noise_model == GPC_NOISE_GAUSSIAN && !getenv("GPC_SPARSE_FULL");
    // Round 4: 64 < capacity <= 120 (the reference's default is 100, /root/reference/src/sparse_gp.h:48): the lower triangles of C and Q
    // of the patch in flight stay in LDS, packed -- four waves per patch, one patch per CU (sparse_add_kernel<.., RES>; GPC_SPARSE_NO_RES=1
    // keeps the HBM-resident triangular mode of the two-wave shape)
    const size_t lds_res = sp_add_lds(g->ld, true) + sizeof(double) * (size_t)(cap_ + 1) * (size_t)(cap_ + 2);
    // MEASURED AND NOT THE DEFAULT (GPC_SPARSE_RES=1 selects it; 8192 patches x 256 points, basis-filling kernel, same box): capacity 100:
    // 55 k patches/s against 111 k for the HBM-resident two-wave shape, capacity 120: 54 k against 83 k, capacity 80 (two workgroups per
    // CU fit): 125 k against 141 k.  A point costs ~16 us of one workgroup either way (the pass is ~40 instructions per element and the
    // ~20 barrier-separated phases of a point do not overlap inside ONE workgroup), and with the state in LDS a CU hosts one patch where
    // the HBM-resident shape hosts four: the stream it removes was already hidden.  States bit-identical to the HBM-resident mode
    // (tests/test_sparse_gpu.py::test_sparse_lds_resident_mode_is_bit_identical).
    const bool res = tri && cap_ > 64 && lds_res <= 160u * 1024u && getenv("GPC_SPARSE_RES") && !getenv("GPC_SPARSE_NO_RES") && !getenv("GPC_SPARSE_WIDE");
    const int nth = res ? SP_THREADS : nth0;
    const size_t lds = res ? lds_res : sp_add_lds(g->ld, tri);
    A.tri_min = 32;   // (measured at the C4 size: 32 -> 62.0 k patches/s, 96 -> 59.0 k, 160 -> 51.1 k; full passes 42.0 k; the defaults regime is level)
    if (const char* e = getenv("GPC_SPARSE_TRI_MIN")) A.tri_min = atoi(e);      // (diagnostic: where the triangular passes start to pay)
    int per_cu = (int)((160u * 1024u) / lds);
    const int per_cu_max = 2 * SP_THREADS / nth;
    per_cu = per_cu > per_cu_max ? per_cu_max : (per_cu < 1 ? 1 : per_cu);
    if (const char* e = getenv("GPC_SPARSE_PER_CU")) per_cu = std::max(1, atoi(e));   // diagnostic: resident-state experiments
    int grid = std::min(g->P, ctx->num_cus * per_cu);
    A.start_it = nullptr;
    A.done_it = nullptr;
    A.list = nullptr;
    A.list_n = nullptr;
    A.out_list = nullptr;
    A.out_list_n = nullptr;
    A.ticket_slot = 0;
    const bool gauss = A.prm.noise_model == GPC_NOISE_GAUSSIAN;
    // three work lists in a row, then their counters (4 each): a phase draws from one and appends what it hands on to the next
    int32_t* const L[3] = {g->list, g->list + g->P, g->list + 2 * (size_t)g->P};
    int32_t* const N[3] = {g->list + 3 * (size_t)g->P, g->list + 3 * (size_t)g->P + 4, g->list + 3 * (size_t)g->P + 8};
    int cur = 0;                                   // the list the next phase draws from
    if (!getenv("GPC_SPARSE_NO_SMALL")) {
        // rows phase first (Gaussian noise): four patches per wave while a patch needs at most 16 basis vectors and no deletion;
        // what it does not finish goes on the work list of the phases below
        if (gauss && !getenv("GPC_SPARSE_NO_ROWS")) {
            constexpr int G = 16, R = 64 / G;
            const size_t lds_r = sizeof(double) * (size_t)(64 + R * (2 * G * G + 4 * G + 16));
            int per_cu_r = std::min(8, (int)((160u * 1024u) / lds_r));
            if (const char* e = getenv("GPC_SPARSE_ROWS_PER_CU")) per_cu_r = std::max(1, std::min(per_cu_r, atoi(e)));   // diagnostic
            A.done_it = g->done_it;
            if (!getenv("GPC_SPARSE_NO_LIST")) {
                A.list = L[0];
                A.list_n = N[0];
                GPC_HIP(ctx, hipMemsetAsync(N[0], 0, 12 * sizeof(int32_t), ctx->stream));
            }
            const int waves = (g->P + R - 1) / R;
            // (a wave per four patches, placed by the hardware as slots come free: patches that leave the phase early end their wave early,
            // and persistent waves striding over the batch carried that imbalance to the end -- 1.93 -> 1.88 ms for the colour GP's pass,
            // level for the depth GP; GPC_SPARSE_ROWS_PERSISTENT=1: eight waves per CU striding, as before)
            const int grid_r = getenv("GPC_SPARSE_ROWS_PERSISTENT") ? std::min(waves, ctx->num_cus * per_cu_r) : waves;
            if (A.ny == 1) hipLaunchKernelGGL((sparse_add_rows_kernel<G, 1>), dim3(grid_r), dim3(64), lds_r, ctx->stream, A);
            else hipLaunchKernelGGL((sparse_add_rows_kernel<G, 3>), dim3(grid_r), dim3(64), lds_r, ctx->stream, A);
            GPC_HIP(ctx, hipGetLastError());
            A.start_it = g->done_it;
        }
        // SECOND rows phase (Gaussian noise, round 4): the patches of the list with at most SP_BMAX vectors, two per wave by ticket, until
        // they outgrow SP_BMAX or ask for a deletion, in place of the one-wave kernel below -- a third of that kernel's instructions per
        // point, and a chain of points is as long as the instructions of its steps: 0.36 against 0.51 ms per add call at the reference's
        // defaults (with the first entry of a group taken by index, not by ticket: 4096 tickets at launch were 0.1 ms), bit-identical
        // states.  A patch that asks for a geometric deletion leaves for the mid phase.  GPC_SPARSE_NO_ROWS2=1: the one-wave kernel.
        bool rows2 = false;
        if (gauss && A.list && A.start_it && !getenv("GPC_SPARSE_NO_ROWS2")) {
            constexpr int G2 = 32, B2 = SP_BMAX, R2 = 64 / G2;
            static_assert(B2 == 24, "the second rows phase is built for a resident block of 24");
            const size_t lds_2 = sizeof(double) * (size_t)(64 + R2 * (2 * B2 * B2 + 4 * B2));
            const int per_cu_2 = std::min(8, (int)((160u * 1024u) / lds_2));
            A.ticket_slot = 1;
            A.out_list = L[cur + 1];
            A.out_list_n = N[cur + 1];
            const int waves2 = (g->P + R2 - 1) / R2;
            if (A.ny == 1) hipLaunchKernelGGL((sparse_add_rows_kernel<G2, 1, B2, true>), dim3(std::min(waves2, ctx->num_cus * per_cu_2)), dim3(64), lds_2, ctx->stream, A);
            else hipLaunchKernelGGL((sparse_add_rows_kernel<G2, 3, B2, true>), dim3(std::min(waves2, ctx->num_cus * per_cu_2)), dim3(64), lds_2, ctx->stream, A);
            GPC_HIP(ctx, hipGetLastError());
            rows2 = true;
            ++cur;
            A.list = L[cur];
            A.list_n = N[cur];
            A.out_list = A.out_list_n = nullptr;
        }
        // small-basis phase: one wave per patch, C and Q in LDS, until a patch outgrows SP_BMAX basis vectors
        const size_t lds_s = sp_add_lds_small();
        int per_cu_s = (int)((160u * 1024u) / lds_s);
        per_cu_s = per_cu_s > 16 ? 16 : per_cu_s;
        if (const char* e = getenv("GPC_SPARSE_SMALL_PER_CU")) per_cu_s = std::max(1, std::min(per_cu_s, atoi(e)));   // diagnostic: occupancy experiments
        A.done_it = g->done_it;
        A.ticket_slot = 1;
        if (!rows2 && A.list) {
            A.out_list = L[cur + 1];
            A.out_list_n = N[cur + 1];
        }
        if (!gauss)
            hipLaunchKernelGGL((sparse_add_kernel<true, true>), dim3(std::min(g->P, ctx->num_cus * per_cu_s)), dim3(64), lds_s, ctx->stream, A);
        else if (!rows2)
            hipLaunchKernelGGL((sparse_add_kernel<true, false>), dim3(std::min(g->P, ctx->num_cus * per_cu_s)), dim3(64), lds_s, ctx->stream, A);
        GPC_HIP(ctx, hipGetLastError());
        if (!rows2 && A.list) {
            ++cur;
            A.list = L[cur];
            A.list_n = N[cur];
            A.out_list = A.out_list_n = nullptr;
        }
        A.start_it = g->done_it;
        // mid phase (Gaussian noise, round 4): the patches that have outgrown SP_BMAX vectors -- a few hundred of 32768 at the reference's
        // defaults, each a chain of up to n points -- go through a second instance of the one-wave kernel with a resident block of
        // SP_BMID before the regular kernel sees them: their state in LDS instead of round trips to HBM at every point
        // (GPC_SPARSE_NO_MID=1: straight to the regular kernel)
        if (gauss && A.list && !getenv("GPC_SPARSE_NO_MID") && g->ld > SP_BMAX) {
            const size_t lds_m = sp_add_lds_small(SP_BMID);
            const int per_cu_m = std::max(1, (int)((160u * 1024u) / lds_m));
            A.ticket_slot = 1;
            A.out_list = L[cur + 1];
            A.out_list_n = N[cur + 1];
            hipLaunchKernelGGL((sparse_add_kernel<true, false, false, false, SP_BMID>), dim3(std::min(g->P, ctx->num_cus * per_cu_m)), dim3(64), lds_m,
                               ctx->stream, A);
            GPC_HIP(ctx, hipGetLastError());
            ++cur;
            A.list = L[cur];
            A.list_n = N[cur];
            A.out_list = A.out_list_n = nullptr;
        }
        A.ticket_slot = 1;
        A.done_it = nullptr;
    }
    if (!gauss) hipLaunchKernelGGL((sparse_add_kernel<false, true>), dim3(grid), dim3(nth), lds, ctx->stream, A);
    else if (res) {
        // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
        GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_add_kernel<false, false, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((sparse_add_kernel<false, false, true, true>), dim3(grid), dim3(nth), lds, ctx->stream, A);
    } else if (tri) hipLaunchKernelGGL((sparse_add_kernel<false, false, true>), dim3(grid), dim3(nth), lds, ctx->stream, A);
    else hipLaunchKernelGGL((sparse_add_kernel<false, false>), dim3(grid), dim3(nth), lds, ctx->stream, A);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

int gpc_sparse_predict_dev(gpc_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma,
                           int conf, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (m < 0) return gpc_fail(ctx, GPC_EINVAL, "negative m");
    if (m > 0 && (!xs0 || !xs1 || !f_star)) return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1/f_star is NULL");
    if (g->P == 0 || m == 0) return GPC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    SpPredParams A;
    A.prm = g->prm;
    A.c_exp = (double)(-0.5f) / g->prm.l_sq;
    A.P = g->P; A.ny = g->ny; A.ld = g->ld; A.m = m; A.conf = conf;
    A.xs0 = xs0; A.xs1 = xs1; A.alpha = g->alpha; A.C = g->C; A.BV = g->BV; A.b = g->b;
    A.f_star = f_star; A.sigma = sigma; A.status_out = status; A.stat = g->stat;
    A.off = nullptr; A.n_total = 0;
    A.fast = (sigma != nullptr && sp_pred_lds(g->ld, true, true) <= 160u * 1024u) ? 1 : 0;
    const size_t lds = sp_pred_lds(g->ld, sigma != nullptr, A.fast != 0);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_predict_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int per_cu = (int)((160u * 1024u) / lds);
    { const int cap_blocks = sigma ? 4 : 8;   // mean only: a patch is a few hundred kernel evaluations, more resident blocks hide their latency
      per_cu = per_cu > cap_blocks ? cap_blocks : (per_cu < 1 ? 1 : per_cu); }
    int grid = std::min(g->P, ctx->num_cus * per_cu);
    return sp_predict_launch(ctx, A, grid, lds);
}

// predict_measurements on every patch's OWN point set (ragged, like the add call's batch): the reference's per-patch training-set
// RMS block (/root/reference/src/gp_compressor.cpp:303-315) calls gps[i].predict_measurements(f, X_i, sigma) exactly so.
int gpc_sparse_predict_points_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1,
                                  double* f, double* sigma, int conf, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (n_total < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (n_total > 0 && (!x0 || !x1 || !f)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/f is NULL");
    if (g->P == 0) return GPC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    SpPredParams A;
    A.prm = g->prm;
    A.c_exp = (double)(-0.5f) / g->prm.l_sq;
    A.P = g->P; A.ny = g->ny; A.ld = g->ld; A.m = 0; A.conf = conf;
    A.xs0 = x0; A.xs1 = x1; A.alpha = g->alpha; A.C = g->C; A.BV = g->BV; A.b = g->b;
    A.f_star = f; A.sigma = sigma; A.status_out = status; A.stat = g->stat;
    A.off = off; A.n_total = n_total;
    A.fast = (sigma != nullptr && sp_pred_lds(g->ld, true, true) <= 160u * 1024u) ? 1 : 0;
    const size_t lds = sp_pred_lds(g->ld, sigma != nullptr, A.fast != 0);
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_predict_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int per_cu = (int)((160u * 1024u) / lds);
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    int grid = std::min(g->P, ctx->num_cus * per_cu);
    return sp_predict_launch(ctx, A, grid, lds);
}

// caller holds ctx->mu.  raw != nullptr: the train_sigmaf pass (sigma_f^2 = 1, per-point sums only)
static int sp_likelihood_launch(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1, const double* y,
                                double* dX, double* l, double* raw)
{
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    SpLikParams A;
    A.prm = g->prm;
    A.raw = raw;
    if (raw) A.prm.sigmaf_sq = 1.0;
    A.c_exp = (double)(-0.5f) / g->prm.l_sq;
    A.P = g->P; A.ny = g->ny; A.ld = g->ld; A.n_total = n_total;
    A.off = off; A.x0 = x0; A.x1 = x1; A.y = y;
    A.alpha = g->alpha; A.C = g->C; A.BV = g->BV; A.b = g->b;
    A.dX = dX; A.l = l;
    A.fast = (sp_lik_lds(g->ld, true) <= 160u * 1024u) ? 1 : 0;
    const size_t lds = sp_lik_lds(g->ld, A.fast != 0);
    // per call: the attribute is per device, and a process may hold contexts on several GPUs (idempotent, host-side only)
    GPC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_likelihood_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int per_cu = (int)((160u * 1024u) / lds);
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    int grid = std::min(g->P, ctx->num_cus * per_cu);
    hipLaunchKernelGGL(sparse_likelihood_kernel, dim3(grid), dim3(SP_THREADS), lds, ctx->stream, A);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

int gpc_sparse_likelihood_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1,
                              const double* y, double* dX, double* l)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (n_total < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (n_total > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (g->prm.noise_model != 0) return gpc_fail(ctx, GPC_EINVAL, "likelihoods are defined for the Gaussian noise model");
    if (g->P == 0 || n_total == 0 || (!dX && !l)) return GPC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    return sp_likelihood_launch(g, off, n_total, x0, x1, y, dX, l, nullptr);
}

#define GPC_TRAIN_MAX_COUNTER 10000

int gpc_sparse_train_sigmaf_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1, const double* y,
                                double step, int max_counter, double* p0, int32_t* iters, double* ls, double* delta)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->ny != 1) return gpc_fail(ctx, GPC_EINVAL, "train_parameters exists for sparse_gp (ny == 1) only");
    if (g->prm.noise_model != 0) return gpc_fail(ctx, GPC_EINVAL, "likelihoods are defined for the Gaussian noise model");
    if (g->P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (n_total < 0) return gpc_fail(ctx, GPC_EINVAL, "negative size");
    if (max_counter < 0 || max_counter > GPC_TRAIN_MAX_COUNTER) return gpc_fail(ctx, GPC_EINVAL, "max_counter must be in [0, %d]", GPC_TRAIN_MAX_COUNTER);
    if (!(step == step)) return gpc_fail(ctx, GPC_EINVAL, "step is NaN");
    if (n_total > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (g->P > 0 && (!p0 || !iters || !ls || !delta)) return gpc_fail(ctx, GPC_EINVAL, "p0/iters/ls/delta is NULL");
    if (g->P == 0) return GPC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcp = gpc_debug_poison_lds(ctx)) return rcp;
    int rc = gpc_ws_reserve(ctx, sizeof(double) * 3 * (size_t)(n_total > 0 ? n_total : 1));
    if (rc != GPC_OK) return rc;
    double* raw = static_cast<double*>(ctx->ws);
    if (n_total > 0) {
        rc = sp_likelihood_launch(g, off, n_total, x0, x1, y, nullptr, nullptr, raw);
        if (rc != GPC_OK) return rc;
    }
    SpTrainParams T;
    T.P = g->P; T.max_counter = max_counter;
    T.sf = g->prm.sigmaf_sq; T.l_sq = g->prm.l_sq; T.s20 = g->prm.noise; T.step = step;
    T.off = off; T.b = g->b; T.raw = raw; T.y = y;
    T.p0 = p0; T.ls = ls; T.delta = delta; T.iters = iters;
    const int wpb = SP_THREADS / 64;
    hipLaunchKernelGGL(sparse_train_kernel, dim3((g->P + wpb - 1) / wpb), dim3(SP_THREADS), 0, ctx->stream, T);
    GPC_HIP(ctx, hipGetLastError());
    return GPC_OK;
}

int gpc_sparse_train_sigmaf(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y, double step,
                            int max_counter, double* p0, int32_t* iters, double* ls, double* delta)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    const int P = g->P;
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (P == 0) return GPC_OK;
    if (off[0] != 0) return gpc_fail(ctx, GPC_EINVAL, "off[0] must be 0");
    for (int i = 0; i < P; ++i)
        if (off[i + 1] < off[i]) return gpc_fail(ctx, GPC_EINVAL, "off must be non-decreasing (patch %d)", i);
    if (max_counter < 0 || max_counter > GPC_TRAIN_MAX_COUNTER) return gpc_fail(ctx, GPC_EINVAL, "max_counter must be in [0, %d]", GPC_TRAIN_MAX_COUNTER);
    const size_t N = (size_t)off[P], Pz = (size_t)P, W = (size_t)max_counter + 2;
    if (N > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (!p0 || !iters || !ls || !delta) return gpc_fail(ctx, GPC_EINVAL, "p0/iters/ls/delta is NULL");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_p0 = nullptr, *d_it = nullptr, *d_ls = nullptr, *d_de = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_off, d_x0, d_x1, d_y, d_p0, d_it, d_ls, d_de})
            if (p) (void)hipFree(p);
    };
    hipStream_t s = gpc_stream_of(ctx);
    hipError_t e = hipMalloc(&d_off, 4 * (Pz + 1));
    if (e == hipSuccess) e = hipMalloc(&d_x0, 8 * (N + 1));
    if (e == hipSuccess) e = hipMalloc(&d_x1, 8 * (N + 1));
    if (e == hipSuccess) e = hipMalloc(&d_y, 8 * (N + 1));
    if (e == hipSuccess) e = hipMalloc(&d_p0, 8 * Pz);
    if (e == hipSuccess) e = hipMalloc(&d_it, 4 * Pz);
    if (e == hipSuccess) e = hipMalloc(&d_ls, 8 * Pz * W);
    if (e == hipSuccess) e = hipMalloc(&d_de, 16 * Pz);
    if (e == hipSuccess) e = hipMemsetAsync(d_ls, 0, 8 * Pz * W, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, 4 * (Pz + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x0, x0, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x1, x1, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_y, y, 8 * N, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_train_sigmaf: %s", hipGetErrorString(e));
    }
    int rc = gpc_sparse_train_sigmaf_dev(g, (const int32_t*)d_off, (int)N, (const double*)d_x0, (const double*)d_x1, (const double*)d_y,
                                         step, max_counter, (double*)d_p0, (int32_t*)d_it, (double*)d_ls, (double*)d_de);
    if (rc == GPC_OK) e = hipMemcpyAsync(p0, d_p0, 8 * Pz, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess) e = hipMemcpyAsync(iters, d_it, 4 * Pz, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess) e = hipMemcpyAsync(ls, d_ls, 8 * Pz * W, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess) e = hipMemcpyAsync(delta, d_de, 16 * Pz, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return gpc_fail(ctx, GPC_EHIP, "gpc_sparse_train_sigmaf: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

int gpc_sparse_likelihood(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y,
                          double* dX, double* l)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    const int P = g->P;
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (P == 0) return GPC_OK;
    if (off[0] != 0) return gpc_fail(ctx, GPC_EINVAL, "off[0] must be 0");
    for (int i = 0; i < P; ++i)
        if (off[i + 1] < off[i]) return gpc_fail(ctx, GPC_EINVAL, "off must be non-decreasing (patch %d)", i);
    const size_t N = (size_t)off[P];
    if (N > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (N == 0 || (!dX && !l)) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_dX = nullptr, *d_l = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_off, d_x0, d_x1, d_y, d_dX, d_l})
            if (p) (void)hipFree(p);
    };
    hipStream_t s = gpc_stream_of(ctx);
    hipError_t e = hipMalloc(&d_off, 4 * (size_t)(P + 1));
    if (e == hipSuccess) e = hipMalloc(&d_x0, 8 * N);
    if (e == hipSuccess) e = hipMalloc(&d_x1, 8 * N);
    if (e == hipSuccess) e = hipMalloc(&d_y, 8 * N * g->ny);
    if (e == hipSuccess && dX) e = hipMalloc(&d_dX, 8 * N * 3);
    if (e == hipSuccess && l) e = hipMalloc(&d_l, 8 * N);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, 4 * (size_t)(P + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_x0, x0, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_x1, x1, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_y, y, 8 * N * g->ny, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_likelihood: %s", hipGetErrorString(e));
    }
    int rc = gpc_sparse_likelihood_dev(g, (const int32_t*)d_off, (int)N, (const double*)d_x0, (const double*)d_x1,
                                       (const double*)d_y, (double*)d_dX, (double*)d_l);
    if (rc == GPC_OK && dX) e = hipMemcpyAsync(dX, d_dX, 8 * N * 3, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess && l) e = hipMemcpyAsync(l, d_l, 8 * N, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return gpc_fail(ctx, GPC_EHIP, "gpc_sparse_likelihood: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

int gpc_sparse_add(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y,
                   const int32_t* perm, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    const int P = g->P;
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (P == 0) return GPC_OK;
    if (off[0] != 0) return gpc_fail(ctx, GPC_EINVAL, "off[0] must be 0");
    int n_max = 0;
    for (int i = 0; i < P; ++i) {
        const int n = off[i + 1] - off[i];
        if (n < 0) return gpc_fail(ctx, GPC_EINVAL, "off must be non-decreasing (patch %d)", i);
        n_max = std::max(n_max, n);
    }
    const size_t N = (size_t)off[P];
    if (N > 0 && (!x0 || !x1 || !y)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/y is NULL");
    if (perm)
        for (int i = 0; i < P; ++i)
            for (int k = off[i]; k < off[i + 1]; ++k)
                if (perm[k] < 0 || perm[k] >= off[i + 1] - off[i])
                    return gpc_fail(ctx, GPC_EINVAL, "perm[%d] = %d outside patch %d", k, perm[k], i);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_y = nullptr, *d_perm = nullptr, *d_st = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_off, d_x0, d_x1, d_y, d_perm, d_st})
            if (p) (void)hipFree(p);
    };
    hipStream_t s = gpc_stream_of(ctx);
    hipError_t e = hipMalloc(&d_off, 4 * (size_t)(P + 1));
    if (e == hipSuccess) e = hipMalloc(&d_x0, 8 * std::max<size_t>(N, 1));
    if (e == hipSuccess) e = hipMalloc(&d_x1, 8 * std::max<size_t>(N, 1));
    if (e == hipSuccess) e = hipMalloc(&d_y, 8 * std::max<size_t>(N, 1) * g->ny);
    if (e == hipSuccess && perm) e = hipMalloc(&d_perm, 4 * std::max<size_t>(N, 1));
    if (e == hipSuccess) e = hipMalloc(&d_st, 4 * (size_t)P);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, 4 * (size_t)(P + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x0, x0, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x1, x1, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_y, y, 8 * N * g->ny, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N && perm) e = hipMemcpyAsync(d_perm, perm, 4 * N, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_add: %s", hipGetErrorString(e));
    }
    int rc = gpc_sparse_add_dev(g, (const int32_t*)d_off, n_max, (int)N, (const double*)d_x0, (const double*)d_x1,
                                (const double*)d_y, (const int32_t*)d_perm, (int32_t*)d_st);
    if (rc == GPC_OK && status) e = hipMemcpyAsync(status, d_st, 4 * (size_t)P, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return gpc_fail(ctx, GPC_EHIP, "gpc_sparse_add: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

int gpc_sparse_predict(gpc_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma,
                       int conf, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (m < 0) return gpc_fail(ctx, GPC_EINVAL, "negative m");
    if (m > 0 && (!xs0 || !xs1 || !f_star)) return gpc_fail(ctx, GPC_EINVAL, "xs0/xs1/f_star is NULL");
    const int P = g->P;
    if (P == 0 || m == 0) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_xs0 = nullptr, *d_xs1 = nullptr, *d_f = nullptr, *d_s = nullptr, *d_st = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_xs0, d_xs1, d_f, d_s, d_st})
            if (p) (void)hipFree(p);
    };
    hipStream_t s = gpc_stream_of(ctx);
    const size_t fbytes = 8 * (size_t)P * g->ny * m, sbytes = 8 * (size_t)P * m;
    hipError_t e = hipMalloc(&d_xs0, 8 * (size_t)m);
    if (e == hipSuccess) e = hipMalloc(&d_xs1, 8 * (size_t)m);
    if (e == hipSuccess) e = hipMalloc(&d_f, fbytes);
    if (e == hipSuccess && sigma) e = hipMalloc(&d_s, sbytes);
    if (e == hipSuccess) e = hipMalloc(&d_st, 4 * (size_t)P);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xs0, xs0, 8 * (size_t)m, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xs1, xs1, 8 * (size_t)m, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_predict: %s", hipGetErrorString(e));
    }
    int rc = gpc_sparse_predict_dev(g, m, (const double*)d_xs0, (const double*)d_xs1, (double*)d_f, (double*)d_s, conf,
                                    (int32_t*)d_st);
    if (rc == GPC_OK) e = hipMemcpyAsync(f_star, d_f, fbytes, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess && sigma) e = hipMemcpyAsync(sigma, d_s, sbytes, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess && status) e = hipMemcpyAsync(status, d_st, 4 * (size_t)P, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return gpc_fail(ctx, GPC_EHIP, "gpc_sparse_predict: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

int gpc_sparse_predict_points(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, double* f, double* sigma,
                              int conf, int32_t* status)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    const int P = g->P;
    if (P > 0 && !off) return gpc_fail(ctx, GPC_EINVAL, "off is NULL");
    if (P == 0) return GPC_OK;
    if (off[0] != 0) return gpc_fail(ctx, GPC_EINVAL, "off[0] must be 0");
    for (int i = 0; i < P; ++i)
        if (off[i + 1] < off[i]) return gpc_fail(ctx, GPC_EINVAL, "off must be non-decreasing (patch %d)", i);
    const size_t N = (size_t)off[P];
    if (N > 0 && (!x0 || !x1 || !f)) return gpc_fail(ctx, GPC_EINVAL, "x0/x1/f is NULL");
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    void *d_off = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_f = nullptr, *d_s = nullptr, *d_st = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_off, d_x0, d_x1, d_f, d_s, d_st})
            if (p) (void)hipFree(p);
    };
    hipStream_t s = gpc_stream_of(ctx);
    const size_t N1 = std::max<size_t>(N, 1);
    hipError_t e = hipMalloc(&d_off, 4 * (size_t)(P + 1));
    if (e == hipSuccess) e = hipMalloc(&d_x0, 8 * N1);
    if (e == hipSuccess) e = hipMalloc(&d_x1, 8 * N1);
    if (e == hipSuccess) e = hipMalloc(&d_f, 8 * N1 * g->ny);
    if (e == hipSuccess && sigma) e = hipMalloc(&d_s, 8 * N1);
    if (e == hipSuccess) e = hipMalloc(&d_st, 4 * (size_t)P);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, 4 * (size_t)(P + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x0, x0, 8 * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && N) e = hipMemcpyAsync(d_x1, x1, 8 * N, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        cleanup();
        return gpc_fail(ctx, e == hipErrorOutOfMemory ? GPC_ENOMEM : GPC_EHIP, "gpc_sparse_predict_points: %s", hipGetErrorString(e));
    }
    int rc = gpc_sparse_predict_points_dev(g, (const int32_t*)d_off, (int)N, (const double*)d_x0, (const double*)d_x1, (double*)d_f,
                                           (double*)d_s, conf, (int32_t*)d_st);
    if (rc == GPC_OK && N) e = hipMemcpyAsync(f, d_f, 8 * N * g->ny, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess && sigma && N) e = hipMemcpyAsync(sigma, d_s, 8 * N, hipMemcpyDeviceToHost, s);
    if (rc == GPC_OK && e == hipSuccess && status) e = hipMemcpyAsync(status, d_st, 4 * (size_t)P, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    cleanup();
    if (rc != GPC_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return gpc_fail(ctx, GPC_EHIP, "gpc_sparse_predict_points: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GPC_OK;
}

int gpc_sparse_sizes(gpc_sparse* g, int32_t* bv_count)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (!bv_count) return gpc_fail(ctx, GPC_EINVAL, "bv_count is NULL");
    if (g->P == 0) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = gpc_stream_of(ctx);
    GPC_HIP(ctx, hipMemcpyAsync(bv_count, g->b, 4 * (size_t)g->P, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

int gpc_sparse_get_state(gpc_sparse* g, double* alpha, double* C, double* Q, double* BV)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->P == 0) return GPC_OK;
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ld = (size_t)g->ld, P = (size_t)g->P;
    hipStream_t s = gpc_stream_of(ctx);
    if (alpha) GPC_HIP(ctx, hipMemcpyAsync(alpha, g->alpha, 8 * P * g->ny * ld, hipMemcpyDeviceToHost, s));
    if (C) GPC_HIP(ctx, hipMemcpyAsync(C, g->C, 8 * P * ld * ld, hipMemcpyDeviceToHost, s));
    if (Q) GPC_HIP(ctx, hipMemcpyAsync(Q, g->Q, 8 * P * ld * ld, hipMemcpyDeviceToHost, s));
    if (BV) GPC_HIP(ctx, hipMemcpyAsync(BV, g->BV, 8 * P * ld * 2, hipMemcpyDeviceToHost, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

// Inverse of gpc_sparse_get_state: loads a stored model (the compressed representation of row f3).  alpha and BV are
// required, C and Q may be NULL (zeroed: the mean prediction of the decompressor needs neither; sigma, likelihoods and
// further online growth do).
int gpc_sparse_set_state(gpc_sparse* g, const int32_t* bv_count, const double* alpha, const double* C, const double* Q,
                         const double* BV)
{
    if (!g) return GPC_EINVAL;
    gpc_ctx* ctx = g->ctx;
    if (ctx->dead.load()) return GPC_EINVAL;   // the context went first: the object can only be destroyed (include/gpc.h)
    if (g->P == 0) return GPC_OK;
    if (!bv_count || !alpha || !BV) return gpc_fail(ctx, GPC_EINVAL, "bv_count/alpha/BV is NULL");
    for (int i = 0; i < g->P; ++i)
        if (bv_count[i] < 0 || bv_count[i] > g->ld) return gpc_fail(ctx, GPC_ERANGE, "bv_count[%d] = %d outside [0, %d]", i, bv_count[i], g->ld);
    std::lock_guard<std::mutex> lk(ctx->mu);
    GPC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ld = (size_t)g->ld, P = (size_t)g->P;
    hipStream_t s = ctx->stream;
    GPC_HIP(ctx, hipMemcpyAsync(g->b, bv_count, 4 * P, hipMemcpyHostToDevice, s));
    GPC_HIP(ctx, hipMemcpyAsync(g->count, bv_count, 4 * P, hipMemcpyHostToDevice, s));
    GPC_HIP(ctx, hipMemsetAsync(g->stat, 0, 4 * P, s));
    GPC_HIP(ctx, hipMemcpyAsync(g->alpha, alpha, 8 * P * g->ny * ld, hipMemcpyHostToDevice, s));
    GPC_HIP(ctx, hipMemcpyAsync(g->BV, BV, 8 * P * ld * 2, hipMemcpyHostToDevice, s));
    if (C) GPC_HIP(ctx, hipMemcpyAsync(g->C, C, 8 * P * ld * ld, hipMemcpyHostToDevice, s));
    else GPC_HIP(ctx, hipMemsetAsync(g->C, 0, 8 * P * ld * ld, s));
    if (Q) GPC_HIP(ctx, hipMemcpyAsync(g->Q, Q, 8 * P * ld * ld, hipMemcpyHostToDevice, s));
    else GPC_HIP(ctx, hipMemsetAsync(g->Q, 0, 8 * P * ld * ld, s));
    GPC_HIP(ctx, hipStreamSynchronize(s));
    return GPC_OK;
}

}  // extern "C"
