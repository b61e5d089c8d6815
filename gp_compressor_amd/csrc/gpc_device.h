// gpc_device.h -- device-side helpers shared by the dense and sparse kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#define GPC_WAVE 64

// ---------------------------------------------------------------------------------------------------------
// exp() for the RBF kernel.  Table-driven (Tang): x = (64 e + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^e * T[j] * (1 + p(r)), p of degree 5 (truncation r^6/720 < 3.5e-17).  Max error ~1 ulp, which is
// the same class as glibc's exp the reference calls (src/rbf_kernel.cpp:17); parity is stated as a tolerance.
// The table (64 doubles = 512 B) is staged into LDS once per workgroup by gpc_exp_table_init().
// Compiles for host as well so that the CPU test-suite can check its accuracy without a GPU.
// ---------------------------------------------------------------------------------------------------------
#define GPC_EXP_TABLE_SIZE 64

// one copy per translation unit (no -fgpu-rdc); 512 B
static __constant__ double c_gpc_exp_table[GPC_EXP_TABLE_SIZE] = {
#include "gpc_exp_table.inc"
};
static const double h_gpc_exp_table[GPC_EXP_TABLE_SIZE] = {
#include "gpc_exp_table.inc"
};

__host__ __device__ static inline double gpc_exp_tbl(double x, const double* __restrict__ T)
{
    const double INV_LN2_64 = 92.332482616893656768;        // 64 / ln 2
    const double LN2_64_HI = 0x1.62e42fef80000p-7;           // ln2/64 rounded to 34 bits: n * hi is exact for |n| < 2^19
    const double LN2_64_LO = 0x1.1cf79abc9e3b4p-42;          // ln2/64 - hi
    // clamp: outside [-760, 720] the result is 0 / inf anyway; keeps the int conversion in range. NaN falls through.
    double xc = x < -760.0 ? -760.0 : (x > 720.0 ? 720.0 : x);
    double nd = __builtin_rint(xc * INV_LN2_64);
    int n = (int)nd;
    double r = __builtin_fma(-nd, LN2_64_HI, xc);
    r = __builtin_fma(-nd, LN2_64_LO, r);
    int j = n & (GPC_EXP_TABLE_SIZE - 1);
    int e = n >> 6;
    double p = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = __builtin_fma(r, p, 1.0 / 6.0);
    p = __builtin_fma(r, p, 0.5);
    double r2 = r * r;
    p = __builtin_fma(r2, p, r);          // r + r^2 (1/2 + r/6 + r^2/24 + r^3/120)
    double t = T[j];
    double v = __builtin_fma(t, p, t);
#if defined(__HIP_DEVICE_COMPILE__)
    v = __builtin_amdgcn_ldexp(v, e);     // v_ldexp_f64: correct gradual underflow / overflow to inf
#else
    v = __builtin_ldexp(v, e);
#endif
    return (x != x) ? x : v;
}

// exp(x) for x <= 0 (the RBF exponent): same algorithm without the upper clamp and the NaN select (a NaN argument
// still comes out as NaN: r and therefore p are NaN).
__device__ static inline double gpc_exp_neg(double x, const double* __restrict__ T)
{
    const double INV_LN2_64 = 92.332482616893656768;
    const double LN2_64_HI = 0x1.62e42fef80000p-7;
    const double LN2_64_LO = 0x1.1cf79abc9e3b4p-42;
    const double xc = x < -760.0 ? -760.0 : x;
    const double nd = __builtin_rint(xc * INV_LN2_64);
    const int n = (int)nd;
    double r = __builtin_fma(-nd, LN2_64_HI, xc);
    r = __builtin_fma(-nd, LN2_64_LO, r);
    double p = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = __builtin_fma(r, p, 1.0 / 6.0);
    p = __builtin_fma(r, p, 0.5);
    p = __builtin_fma(r * r, p, r);
    const double t = T[n & (GPC_EXP_TABLE_SIZE - 1)];
    return __builtin_amdgcn_ldexp(__builtin_fma(t, p, t), n >> 6);
}

// exp(x) for -2^-5 <= x <= 0: the degree-7 Taylor polynomial, no table and no range reduction (truncation
// x^8/8! <= 2.3e-17 relative, i.e. below half an ulp; measured <= 1 ulp vs libm, tests/test_capi_cpu.py).  This is the
// regime of a GP patch model whose length scale exceeds the patch (c d^2 = -d^2 / (2 l^2) is tiny): a third of the
// instructions of the table-driven path and no LDS access.  Callers prove the argument range from a bound on the patch
// extent (dense_mfma.hip) and fall back to gpc_exp_neg otherwise.
#define GPC_EXP_SMALL_MAX 0.03125
__host__ __device__ static inline double gpc_exp_small(double x)
{
    double p = __builtin_fma(x, 1.0 / 5040.0, 1.0 / 720.0);
    p = __builtin_fma(x, p, 1.0 / 120.0);
    p = __builtin_fma(x, p, 1.0 / 24.0);
    p = __builtin_fma(x, p, 1.0 / 6.0);
    p = __builtin_fma(x, p, 0.5);
    p = __builtin_fma(x, p, 1.0);
    return __builtin_fma(x, p, 1.0);
}

// exp(-t) for 0 <= t <= 2^-5 (DEG 7) / 0 <= t <= 2^-8 (DEG 5): Taylor polynomials with literal coefficients (truncation t^8/8! <= 2.3e-17,
// t^6/6! <= 4.9e-18: below a tenth of an ulp of the result, which is ~1).  Callers hold coordinates pre-scaled by sqrt(-c), so that the
// squared distance IS t: distance (4 operations) + 5 or 7 FMAs per kernel value, no multiplication by c (one-wave dense kernel, round 4).
#define GPC_EXP_TINY_MAX 0.00390625
template <int DEG>
__host__ __device__ static inline double gpc_expm_poly(double t)
{
    double p;
    if (DEG == 7) {
        p = __builtin_fma(t, -1.0 / 5040.0, 1.0 / 720.0);
        p = __builtin_fma(t, p, -1.0 / 120.0);
        p = __builtin_fma(t, p, 1.0 / 24.0);
    } else {
        p = __builtin_fma(t, -1.0 / 120.0, 1.0 / 24.0);
    }
    p = __builtin_fma(t, p, -1.0 / 6.0);
    p = __builtin_fma(t, p, 0.5);
    p = __builtin_fma(t, p, -1.0);
    return __builtin_fma(t, p, 1.0);
}

__device__ static inline void gpc_exp_table_init(double* T_lds)
{
    for (int i = threadIdx.x; i < GPC_EXP_TABLE_SIZE; i += blockDim.x) T_lds[i] = c_gpc_exp_table[i];
}

// RBF / squared-exponential kernel.  c = (double)(-0.5f) / l_sq is formed on the host exactly like the reference
// evaluates `-0.5f / p(1)` (src/rbf_kernel.cpp:17, src/gaussian_process.cpp:49); sf = sigma_f^2.
__device__ static inline double gpc_rbf(double sf, double c, double xi0, double xi1, double xj0, double xj1, const double* T)
{
    double d0 = xi0 - xj0, d1 = xi1 - xj1;
    double sq = __builtin_fma(d0, d0, d1 * d1);   // explicit: left to the compiler, WHICH product is fused differs from one inlining context to the next
    return sf * gpc_exp_tbl(c * sq, T);
}

// same with the x <= 0 exponential (c < 0 because l_sq > 0 is checked by the API)
__device__ static inline double gpc_rbf_neg(double sf, double c, double xi0, double xi1, double xj0, double xj1, const double* T)
{
    double d0 = xi0 - xj0, d1 = xi1 - xj1;
    double sq = __builtin_fma(d0, d0, d1 * d1);   // explicit: left to the compiler, WHICH product is fused differs from one inlining context to the next
    return sf * gpc_exp_neg(c * sq, T);
}

__device__ static inline double gpc_rbf_small(double sf, double c, double xi0, double xi1, double xj0, double xj1)
{
    double d0 = xi0 - xj0, d1 = xi1 - xj1;
    double sq = __builtin_fma(d0, d0, d1 * d1);   // explicit: left to the compiler, WHICH product is fused differs from one inlining context to the next
    return sf * gpc_exp_small(c * sq);
}

// ---------------------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------------------
__device__ static inline double gpc_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum through LDS scratch (>= blockDim.x/64 doubles); every thread gets the result
__device__ static inline double gpc_block_sum(double v, double* scratch)
{
    v = gpc_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += scratch[i];
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// Noise functors: the duck-typed Noise contract of the reference, q = dx_ln(y, x, sigma_x) = d/dx ln P(y|x) and
// r = dx2_ln(y, x, sigma_x) = d2/dx2 ln P(y|x)   (/root/reference/src/gaussian_noise.h:7-10, src/probit_noise.h).
// One definition for every kernel that needs them (sparse add, the dense IRLS loop, gpc_noise_eval).
// ---------------------------------------------------------------------------------------------------------
#define GPC_NOISE_GAUSSIAN 0
#define GPC_NOISE_PROBIT_REF 1   // probit_noise as written: "Phi"(z) = erf(z) / (2.0f*sqrt(2.0f))   (src/probit_noise.cpp:15,26)
#define GPC_NOISE_PROBIT_STD 2   // the fix: Phi(z) = (1 + erf(z / sqrt 2)) / 2, a CDF; everything else as upstream

// gaussian_noise::dx_ln / dx2_ln   (/root/reference/src/gaussian_noise.cpp:9-18)
__device__ static inline void gpc_gaussian_q_r(double s20, double y, double x, double sigma_x, double* q, double* r)
{
    *q = (y - x) / (s20 + sigma_x);
    *r = (double)(-1.0f) / (s20 + sigma_x);
}

// probit_noise::dx_ln / dx2_ln   (/root/reference/src/probit_noise.cpp:11-31), operation for operation.  `2.0f*sqrt(2.0f)`
// is a float product upstream (the sqrt(float) overload), checked against the compiled reference object in
// tests/golden/noise_ref.json; exp(-z*z/2) and sqrt(2.0f*M_PI) are double.  model == GPC_NOISE_PROBIT_STD swaps the
// one expression that is wrong upstream (ef) and keeps the rest.
__device__ static inline void gpc_probit_q_r(int model, double s20, double y, double x, double sigma_x, double* q, double* r)
{
    const double sigma2 = s20 + sigma_x;
    const double sigma = sqrt(sigma2);
    const double z = y * x / sigma;
    const double ef = (model == GPC_NOISE_PROBIT_STD) ? 0.5 * erfc(-z * 0.70710678118654752440)
                                                      : erf(z) / (double)(2.0f * 1.41421354f);
    const double efprim = exp(-z * z / 2.0) / sqrt(2.0 * 3.14159265358979323846);
    *q = y / sigma * efprim / ef;                                 // :17
    const double efprimprim = -z * efprim;                        // :28
    const double first = efprim / ef;                             // :29
    *r = (efprimprim / ef - first * first) / sigma2;              // :30
}
