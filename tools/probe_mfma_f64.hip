// probe_mfma_f64.hip -- development probe (not part of the product): verifies the v_mfma_f64_16x16x4_f64 operand /
// accumulator lane maps the dense_mfma kernel relies on, and measures the FP64 MFMA / FMA issue rates on this GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_f64.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// D = A(16x4) * B(4x16) + C;  documented maps: A lane l -> A[l&15][l>>4], B lane l -> B[l>>4][l&15],
// D reg r lane l -> D[(l>>4) + 4r][l&15]
__global__ void layout_kernel(const double* A, const double* B, double* D, int blgp)
{
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    if (blgp == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    else if (blgp == 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 1);
    else if (blgp == 2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 2);
    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 4);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate_kernel(double* out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_rate_kernel(double* out, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d kHz, LDS/block %zu, regs/block %d, L2 %d\n", prop.name, prop.multiProcessorCount,
           prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock, prop.l2CacheSize);
    // ---- layout
    std::vector<double> A(64), B(64), D(256), R(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * 10 + k + 1;      // asymmetric
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k + 1) * 100 + 3 * j + 1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    for (int blgp = 0; blgp < 4; ++blgp) {
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, blgp);
        CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
        int bad = 0, neg = 0;
        for (int i = 0; i < 256; ++i) { if (D[i] != R[i]) ++bad; if (D[i] == -R[i]) ++neg; }
        printf("layout blgp=%d: mismatches vs A*B = %d, equal to -(A*B) = %d  (D[0][0]=%g want %g, D[3][5]=%g want %g)\n",
               blgp == 3 ? 4 : blgp, bad, neg, D[0], R[0], D[3 * 16 + 5], R[3 * 16 + 5]);
    }
    // ---- rates
    const int blocks = prop.multiProcessorCount * 2, iters = 20000;
    double* out; CK(hipMalloc(&out, (size_t)blocks * 2 * 256 * 8));   // largest launch below is blocks*2
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double fl = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4  4 acc, %d blocks x 4 waves: %.3f ms  %.2f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
        double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 4 * 2 /*blocks per CU*/);
        printf("   ~%.1f cycles per MFMA per SIMD at 2.4 GHz (2 waves/SIMD)\n", cyc);
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(prop.multiProcessorCount), dim3(256), 0, 0, out, iters); });
        printf("mfma_f64 dependent chain (1 acc, 1 wave/SIMD): %.1f cycles per MFMA at 2.4 GHz\n", ms * 1e-3 * 2.4e9 / iters);
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(prop.multiProcessorCount), dim3(256), 0, 0, out, iters); });
        printf("mfma_f64 2 acc, 1 wave/SIMD: %.1f cycles per MFMA at 2.4 GHz\n", ms * 1e-3 * 2.4e9 / (iters * 2));
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(fma_rate_kernel<8>, dim3(blocks * 2), dim3(256), 0, 0, out, iters); });
        double fl = (double)blocks * 2 * 256 * 8 * (double)iters * 2;
        printf("v_fma_f64 8 acc, %d blocks: %.3f ms  %.2f TFLOP/s\n", blocks * 2, ms, fl / ms / 1e9);
    }
    return 0;
}
