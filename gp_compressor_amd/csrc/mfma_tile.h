// mfma_tile.h -- device helpers shared by the register-tile kernels (dense_mfma.hip, dense_mfma_big.hip):
// the MFMA operand image, the 16 x 16 diagonal-tile factor on the MFMA pipe, LDS flag hand-over, DPP row reductions.
// Lane maps of v_mfma_f64_16x16x4_f64 (tools/probe_mfma_f64.hip): A operand lane l = A[l&15][l>>4], B operand lane l =
// B[l>>4][l&15], C/D register r of lane l = D[(l>>4) + 4r][l&15]; blgp = 1 negates A.
#pragma once

#include "gpc_device.h"

typedef double d4 __attribute__((ext_vector_type(4)));
#define MF_TS 16

// The kernel is fully unrolled over tile slots, and every slot has its own lane-dependent LDS addresses.  hipcc
// hoists such loop-invariant address arithmetic out of the surrounding loops and keeps it in registers.  Passing the
// lane id through an empty asm inside each slot body makes the addresses cheap-to-recompute values the compiler
// cannot hoist: one or two extra VALU ops per use instead of a register each.
__device__ static __forceinline__ int mf_opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Spin until *word >= k (an LDS word published with release semantics by another wave of the workgroup).
// Written as ONE asm statement on purpose: as a C loop it puts a cycle into the CFG of the fully unrolled step body
// and hipcc's register allocator answers with ~100 spilled accumulator registers.  All lanes read the same word.
__device__ static __forceinline__ bool mf_wait_ge(unsigned lds_byte_addr, int k)
{
    // Bounded: 2^20 polls x s_sleep(1) is tens of milliseconds, three orders of magnitude beyond any legitimate wait.
    // On expiry the wave simply carries on (every other wait is bounded too, so the workgroup drains) and the patch is
    // reported as GPC_STATUS_NAN: a protocol failure must never hang the GPU.
    int v, cnt;
    asm volatile(
        "s_mov_b32 %1, 0x100000\n\t"
        "1:\n\t"
        "ds_read_b32 %0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_gt_i32 vcc, %3, %0\n\t"
        "s_cbranch_vccz 2f\n\t"
        "s_sub_u32 %1, %1, 1\n\t"
        "s_cmp_eq_u32 %1, 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        : "=&v"(v), "=&s"(cnt)
        : "v"(lds_byte_addr), "v"(k)
        : "vcc", "scc", "memory");
    return cnt != 0;
}
__device__ static __forceinline__ void mf_publish(int* word, int k)
{
    __hip_atomic_store(word, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ static __forceinline__ double mf_readlane(double v, int lane_const)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_const);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_const);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) and 1/d to fp64 accuracy: hardware seed + two Newton steps
__device__ static __forceinline__ double mf_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    return y;
}
__device__ static __forceinline__ double mf_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// ---- operand image ------------------------------------------------------------------------------------------
// A 16 x 16 matrix M as MFMA A/B operand data: lane l needs M[l & 15][(l >> 4) + 4 s], s = 0..3.  The image is two
// planes of 64 x 16 bytes: plane s>>1 holds, for lane l, the pair (s&1 = 0, 1) at byte l*16.  Each ds_read_b128 /
// ds_write_b128 then moves 16 contiguous bytes per lane at a 16-byte lane stride, which is bank-conflict free (a
// single 32-byte-per-lane image makes every b128 access a 2-way conflict: rocprof showed 30 % of the LDS cycles lost).
typedef double d2 __attribute__((ext_vector_type(2)));
#define MF_IMG 256   // doubles per image
__device__ static __forceinline__ int mf_img_off(int l, int s) { return (s >> 1) * 128 + l * 2 + (s & 1); }
// element (r, c) of M (c is the contraction index)
__device__ static __forceinline__ int mf_img_rc(int r, int c) { return mf_img_off(r + 16 * (c & 3), c >> 2); }
__device__ static __forceinline__ d4 mf_img_load(const double* img, int l)
{
    const d2 a = *reinterpret_cast<const d2*>(img + l * 2);
    const d2 b = *reinterpret_cast<const d2*>(img + 128 + l * 2);
    return d4{a[0], a[1], b[0], b[1]};
}
__device__ static __forceinline__ void mf_img_store(double* img, int l, d4 v)
{
    *reinterpret_cast<d2*>(img + l * 2) = d2{v[0], v[1]};
    *reinterpret_cast<d2*>(img + 128 + l * 2) = d2{v[2], v[3]};
}

// ... with the non-temporal hint (a stream that is read or written once should not push re-used lines out of L2)
__device__ static __forceinline__ d4 mf_img_load_nt(const double* img, int l)
{
    const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(img + l * 2));
    const d2 b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(img + 128 + l * 2));
    return d4{a[0], a[1], b[0], b[1]};
}
__device__ static __forceinline__ void mf_img_store_nt(double* img, int l, d4 v)
{
    __builtin_nontemporal_store(d2{v[0], v[1]}, reinterpret_cast<d2*>(img + l * 2));
    __builtin_nontemporal_store(d2{v[2], v[3]}, reinterpret_cast<d2*>(img + 128 + l * 2));
}

// Inverse Cholesky factor of a 16 x 16 SPD tile, on the MFMA pipe.  `W` holds the tile in C/D register layout
// (lane l, register r: A[(l>>4) + 4 r][l & 15]; the diagonal tile is symmetric, so the workers' transposed storage is
// the same thing).  Square-root-free Gauss-Jordan elimination IN PLACE: pivot c is ONE rank-1 v_mfma_f64_16x16x4
//     W[i][j] -= m_i * w_j,   m_i = W[i][c] / p_c (i > c, else 0),   w_j = W[c][j] + [j == c]
// whose only non-zero contraction slot is k = c & 3 -- pivot row c lives in register c >> 2 of exactly the 16 lanes
// that form slot k of both operands, so no data moves across lanes.  Columns j > c are the Schur complement, columns
// j < c the rows of the unit-lower inverse Lu^-1, and the dead column c receives -m_i = Lu^-1[i][c]: after 15 pivots the
// tile is [D Lu^T \ Lu^-1] and L^-1 = D^-1/2 Lu^-1.  The reciprocal of pivot c+1 is formed one step AHEAD from two
// scalars of the current tile (p_(c+1) = W[c+1][c+1] - W[c][c+1]^2 / p_c), so the dependency chain per pivot is
// one MFMA plus two VALU ops; the square roots are taken once, vectorised, at the end.  (The previous version ran the
// elimination on the VALU with v_readlane multipliers: ~650 dependent-issue instructions, 4.6k cycles per tile, on
// the critical path of every step.)  Writes L^-1 and L^-T as operand images; false when a pivot is <= pivot_tol.
template <bool OPAQUE = false>
__device__ __forceinline__ static bool mf_diag_factor(d4 W, double* rsbuf, double* Linv_out, double* LinvT_out, double pivot_tol, int lane_in = -1)
{
    // (OPAQUE: the first pivot's masks too -- hoisted out of the caller's loops they are spilled and every reload waits on vmcnt)
    // (lane_in: a caller that keeps its lane id in a register anyway passes it -- threadIdx.x itself gets spilled around these calls)
    const int lane = OPAQUE ? mf_opaque(lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63)) : (int)(threadIdx.x & 63);
    const int lr = lane & 15, lg = lane >> 4;
    double rp = mf_rcp(mf_readlane(W[0], 0));
    double rpv = (lg == 0 && lr > 0) ? rp : 0.0;
    // The lane masks of a pivot (is this lane in contraction slot q, is it the pivot's own column) are loop-invariant constants of
    // the CALLER's loops.  In the register-tile kernel (dense_mfma.hip) hipcc hoists them and keeps them in registers: fine.  In the
    // tiled kernel it hoists all 15 x 2 of them out of the step loop, runs out of registers and reloads them from scratch in front of
    // every MFMA of this chain: OPAQUE re-derives them from an opaque copy of the lane id instead, in the shadow of the previous
    // pivot's MFMA (a handful of VALU ops; 4 % slower than the hoisted form where that one stays in registers).
    double one_q = (lg == 0) ? 1.0 : 0.0, ev = (lane == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < MF_TS - 1; ++c) {
        const int q = c & 3, r = c >> 2, q1 = (c + 1) & 3, r1 = (c + 1) >> 2;
        if constexpr (!OPAQUE) {
            one_q = (lg == q) ? 1.0 : 0.0;
            ev = (lg == q && lr == c) ? 1.0 : 0.0;
        }
        // on the chain: two VALU ops and the MFMA
        const double a_op = W[r] * rpv;
        const double b_op = __builtin_fma(W[r], one_q, ev);
        const d4 Wn = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op, W, 0, 0, 1);   // blgp = 1: NEG(A)
        __builtin_amdgcn_sched_barrier(0);
        // in the shadow of the MFMA: the reciprocal of the next pivot from the OLD tile (kept alive in its own registers)
        const double s01 = mf_readlane(W[r], 16 * q + c + 1);      // W[c][c+1]
        const double s11 = mf_readlane(W[r1], 16 * q1 + c + 1);    // W[c+1][c+1]
        const double t = s01 * rp;
        rp = mf_rcp(__builtin_fma(-t, s01, s11));
        if constexpr (OPAQUE) {
            const int lo_ = mf_opaque(lane);
            const int lr_ = lo_ & 15, lg_ = lo_ >> 4;
            rpv = (lg_ == q1 && lr_ > c + 1) ? rp : 0.0;
            one_q = (lg_ == q1) ? 1.0 : 0.0;
            ev = (lo_ == 16 * q1 + c + 1) ? 1.0 : 0.0;
        } else {
            rpv = (lg == q1 && lr > c + 1) ? rp : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
        W = Wn;
    }
    // the pivots are the diagonal of the tile: row lg + 4 r == column lr  <=>  lane 16 (i & 3) + i, register i >> 2
    // (OPAQUE: lane maps again from an opaque copy, see above)
    const int lt_ = OPAQUE ? mf_opaque(lane) : lane;
    const int lrt = lt_ & 15, lgt = lt_ >> 4;
    const int rsel = lrt >> 2;
    const double pd = rsel == 0 ? W[0] : rsel == 1 ? W[1] : rsel == 2 ? W[2] : W[3];
    const bool on_diag = (lrt & 3) == lgt;
    const bool ok = __builtin_amdgcn_ballot_w64(on_diag && !(pd > pivot_tol)) == 0;
    const double rs = mf_rsqrt(on_diag ? pd : 1.0);
    if (on_diag) rsbuf[lrt] = rs;       // one wave: LDS operations execute in program order, no barrier needed
    d4 fin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lgt + 4 * r;
        const double rsr = rsbuf[row];
        fin[r] = (lrt < row) ? W[r] * rsr : (lrt == row ? rsr : 0.0);
    }
    mf_img_store(LinvT_out, lt_, fin);     // C/D registers of L^-1 = operand image of L^-T
#pragma unroll
    for (int r = 0; r < 4; ++r) Linv_out[mf_img_rc(lgt + 4 * r, lrt)] = fin[r];
    return ok;
}

// ---- the same factor with CONSTANT lane masks (round 4; the one-wave kernel) -------------------------------------------------------
// Which lanes form contraction slot q, which lane is the pivot's own column, which rows lie below it: per pivot these are three
// 64-bit lane masks known at compile time.  mf_diag_factor derives them from the lane id (hoisted by the compiler: 30 live values that
// spill in the tiled kernels; OPAQUE: re-derived per pivot, ~12 VALU instructions).  Here they are literals in SGPR pairs -- two
// s_mov_b32 on the scalar unit -- and a select is ONE v_cndmask_b32 per 32-bit half: 4 VALU instructions per pivot instead of ~12,
// nothing to hoist, nothing to spill.
__device__ static __forceinline__ int mf_sel32(int v, unsigned long long mask)      // lanes in `mask` keep v, the others get 0
{
    int r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
    return r;
}
__device__ static __forceinline__ double mf_sel64(double v, unsigned long long mask)
{
    return __hiloint2double(mf_sel32(__double2hiint(v), mask), mf_sel32(__double2loint(v), mask));
}
__device__ static __forceinline__ double mf_one_in(unsigned long long mask, int one_hi)   // 1.0 in the lanes of `mask`, 0.0 elsewhere
{
    return __hiloint2double(mf_sel32(one_hi, mask), 0);
}
__device__ __forceinline__ static bool mf_diag_factor_c(d4 W, double* rsbuf, double* Linv_out, double* LinvT_out, double pivot_tol, int lane_in)
{
    const int one_hi = 0x3FF00000;
    double rp = mf_rcp(mf_readlane(W[0], 0));
    double rpv = mf_sel64(rp, 0xFFFEull);                 // lg == 0 && lr > 0
    double one_q = mf_one_in(0xFFFFull, one_hi);          // lg == 0
    double ev = mf_one_in(0x1ull, one_hi);                // lane 0
#pragma unroll
    for (int c = 0; c < MF_TS - 1; ++c) {
        const int q = c & 3, r = c >> 2, q1 = (c + 1) & 3, r1 = (c + 1) >> 2;
        // on the chain: two VALU ops and the MFMA
        const double a_op = W[r] * rpv;
        const double b_op = __builtin_fma(W[r], one_q, ev);
        const d4 Wn = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op, W, 0, 0, 1);   // blgp = 1: NEG(A)
        __builtin_amdgcn_sched_barrier(0);
        // in the shadow of the MFMA: the reciprocal of the next pivot from the OLD tile (kept alive in its own registers)
        const double s01 = mf_readlane(W[r], 16 * q + c + 1);      // W[c][c+1]
        const double s11 = mf_readlane(W[r1], 16 * q1 + c + 1);    // W[c+1][c+1]
        const double t = s01 * rp;
        rp = mf_rcp(__builtin_fma(-t, s01, s11));
        // lanes of slot q1: 16 q1 .. 16 q1 + 15; rows below pivot c + 1: lr > c + 1
        const unsigned long long slot = 0xFFFFull << (16 * q1);
        const unsigned long long below = ((0xFFFFull << (c + 2)) & 0xFFFFull) << (16 * q1);
        rpv = mf_sel64(rp, below);
        one_q = mf_one_in(slot, one_hi);
        ev = mf_one_in(1ull << (16 * q1 + c + 1), one_hi);
        __builtin_amdgcn_sched_barrier(0);
        W = Wn;
    }
    const int lt_ = mf_opaque(lane_in);
    const int lrt = lt_ & 15, lgt = lt_ >> 4;
    const int rsel = lrt >> 2;
    const double pd = rsel == 0 ? W[0] : rsel == 1 ? W[1] : rsel == 2 ? W[2] : W[3];
    const bool on_diag = (lrt & 3) == lgt;
    const bool ok = __builtin_amdgcn_ballot_w64(on_diag && !(pd > pivot_tol)) == 0;
    const double rs = mf_rsqrt(on_diag ? pd : 1.0);
    if (on_diag) rsbuf[lrt] = rs;       // one wave: LDS operations execute in program order, no barrier needed
    d4 fin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lgt + 4 * r;
        const double rsr = rsbuf[row];
        fin[r] = (lrt < row) ? W[r] * rsr : (lrt == row ? rsr : 0.0);
    }
    if (LinvT_out) mf_img_store(LinvT_out, lt_, fin);     // C/D registers of L^-1 = operand image of L^-T
#pragma unroll
    for (int r = 0; r < 4; ++r) Linv_out[mf_img_rc(lgt + 4 * r, lrt)] = fin[r];
    return ok;
}

// out[mr] = sum_kk M[mr][kk] * v[kk] for a 16 x 16 matrix stored as an operand image, one thread per row
__device__ static __forceinline__ double mf_row_dot(const double* img, int mr, const double* v)
{
    double s_ = 0.0;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const d4 ch = mf_img_load(img, mr + 16 * gq);      // k = gq, gq+4, gq+8, gq+12
#pragma unroll
        for (int s = 0; s < 4; ++s) s_ = __builtin_fma(ch[s], v[gq + 4 * s], s_);
    }
    return s_;
}

template <int CTRL>
__device__ static __forceinline__ double mf_dpp(double v)
{
    // (row rotations: every lane has a source lane, the destination's previous value never shows -- the mov form needs no zeroed destination)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
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

// Sums the four components of x over the 16 lanes of a DPP row in 5 exchange rounds instead of 16: after the xor-8 and
// half-mirror rounds a lane keeps only ONE component, sel = 2 (l>>3 & 1) + (l>>2 & 1), which the two quad rounds finish.
// Every lane returns the row total of its component `sel` (so lanes l&3 == 0 hold one total each: 0, 1, 2, 3 at l&15 =
// 0, 4, 8, 12).  ~40 VALU instructions against ~130 for four independent all-reduces.
__device__ static __forceinline__ double mf_row_reduce4(d4 x, int lr)
{
    const bool hi8 = (lr & 8) != 0, hi4 = (lr & 4) != 0;
    double k0 = hi8 ? x[2] : x[0], k1 = hi8 ? x[3] : x[1];
    const double s0 = hi8 ? x[0] : x[2], s1 = hi8 ? x[1] : x[3];
    k0 += mf_dpp<0x128>(s0);            // row_ror:8  (l <-> l ^ 8)
    k1 += mf_dpp<0x128>(s1);
    double k = hi4 ? k1 : k0;
    const double sd = hi4 ? k0 : k1;
    k += mf_dpp<0x141>(sd);             // row_half_mirror (l <-> 7 - l inside each half row)
    k += mf_dpp<0xB1>(k);               // quad_perm [1,0,3,2]
    k += mf_dpp<0x4E>(k);               // quad_perm [2,3,0,1]
    return k;
}

