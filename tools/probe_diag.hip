// Micro-benchmark of mf_diag_factor (csrc/dense_mfma.hip) in isolation: cycles per 16 x 16 tile for the factor wave alone
// and with a wave streaming MFMAs on the same SIMD (the situation inside dense_mfma_kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igp_compressor_amd/csrc tools/probe_diag.hip -o /tmp/probe_diag
#include "../gp_compressor_amd/csrc/dense_mfma.hip"
#include <cstdio>
#include <cmath>
#include <vector>

__global__ __launch_bounds__(512) void probe_kernel(const double* tile, double* out, unsigned long long* cyc, int reps, int mode)
{
    __shared__ __attribute__((aligned(16))) double sm[256 + 32 + 256 + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 256) sm[tid] = tile[tid];
    __syncthreads();
    if (wave == 7) {
        __builtin_amdgcn_s_setprio(3);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        bool ok = true;
        for (int i = 0; i < reps; ++i) {
            const d4 W = *reinterpret_cast<const d4*>(sm + lane * 4);
            ok &= mf_diag_factor(W, sm + 256, sm + 288, sm + 544, 1e-300);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[blockIdx.x] = t1 - t0; out[512] = ok ? 1.0 : 0.0; }
        for (int i = lane; i < 512; i += 64) out[i] = sm[288 + i];
    } else if ((mode & 1) && wave == 3) {
        // same SIMD as wave 7: MFMA streams like a worker's trailing update.  mode bits 2..3 pick the pattern:
        // 0 = four independent accumulators, 1 = one dependent chain, 2 = dependent chain + s_sleep 1 after every 4
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double x = 1.0 + lane * 1e-3, y = 0.5 - lane * 1e-3;
        const int pat = (mode >> 2) & 3;
        for (int i = 0; i < reps * 12; ++i) {
            if (pat == 0) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                if (pat == 2) __builtin_amdgcn_s_sleep(1);
                if (pat == 3) __builtin_amdgcn_s_sleep(4);
            }
        }
        out[1024 + lane] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if ((mode & 2) && wave != 7) {
        // everyone else keeps the FP64 VALU busy
        double s = lane;
        for (int i = 0; i < reps * 300; ++i) s = __builtin_fma(s, 1.0000001, 0.5);
        out[1100 + tid] = s;
    }
}

int main()
{
    const int reps = 200;
    std::vector<double> A(256), W(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = 0.0025 * std::exp(-0.5 / 9.0 * 1e-4 * (i - j) * (i - j)) + (i == j ? 0.0032 : 0.0);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) W[l * 4 + r] = A[((l >> 4) + 4 * r) * 16 + (l & 15)];
    double *dT, *dO;
    unsigned long long* dC;
    hipMalloc(&dT, 256 * 8); hipMalloc(&dO, 4096 * 8); hipMalloc(&dC, 8 * 64);
    hipMemcpy(dT, W.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int mode : {0, 1, 5, 9, 13, 3}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(512), 0, 0, dT, dO, dC, reps, mode);
            hipDeviceSynchronize();
        }
        unsigned long long c; double ok;
        hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&ok, dO + 512, 8, hipMemcpyDeviceToHost);
        static const char* pats[4] = {"4 independent accumulators", "one dependent chain", "dependent chain + s_sleep 1 per 4", "dependent chain + s_sleep 4 per 4"};
        printf("mode %2d (%s%s%s): %.0f s_memtime ticks per diag factor, ok=%g\n", mode, (mode & 1) ? "MFMA stream on the same SIMD: " : "alone",
               (mode & 1) ? pats[(mode >> 2) & 3] : "", (mode & 2) ? " + FP64 VALU load on all other waves" : "", (double)c / reps, ok);
    }
    // check the result: Linv * A * Linv^T = I
    std::vector<double> o(512);
    hipMemcpy(o.data(), dO, 512 * 8, hipMemcpyDeviceToHost);
    double Li[16][16];
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) Li[r][c] = o[((c >> 2) >> 1) * 128 + (r + 16 * (c & 3)) * 2 + ((c >> 2) & 1)];
    double err = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int a = 0; a < 16; ++a)
                for (int b = 0; b < 16; ++b) s += Li[i][a] * A[a * 16 + b] * Li[j][b];
            err = fmax(err, fabs(s - (i == j)));
        }
    printf("max |Linv A Linv^T - I| = %.3g\n", err);
    return 0;
}
