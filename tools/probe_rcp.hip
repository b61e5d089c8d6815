// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 (how many Newton steps mf_rcp / mf_rsqrt need): max relative error of the raw
// instruction and after one and two Newton steps, over 2^20 arguments spread across the exponent range.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_rcp.hip -o tools/probe_rcp_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y = __builtin_amdgcn_rcp(d);
    o[i] = y;
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    o[n + i] = y;
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    o[2 * n + i] = y;
    double z = __builtin_amdgcn_rsq(d);
    o[3 * n + i] = z;
    double f = __builtin_fma(-d * z, z, 1.0);
    z = __builtin_fma(z * 0.5, f, z);
    o[4 * n + i] = z;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;
        x[i] = std::ldexp(m, (int)(s % 120) - 60);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 5 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double e[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / (long double)x[i], q = 1.0L / sqrtl((long double)x[i]);
        for (int j = 0; j < 3; ++j) e[j] = std::fmax(e[j], (double)fabsl(((long double)o[(size_t)j * n + i] - r) / r));
        for (int j = 3; j < 5; ++j) e[j] = std::fmax(e[j], (double)fabsl(((long double)o[(size_t)j * n + i] - q) / q));
    }
    printf("v_rcp_f64: raw %.3e (2^%.1f), one Newton step %.3e (2^%.1f), two %.3e\n", e[0], std::log2(e[0]), e[1], std::log2(e[1]), e[2]);
    printf("v_rsq_f64: raw %.3e (2^%.1f), one Newton step %.3e (2^%.1f)\n", e[3], std::log2(e[3]), e[4], std::log2(e[4]));
    return 0;
}
