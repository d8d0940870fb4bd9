// clock_after_mfma.hip - does a latency-bound VALU kernel run slower right after dense fp64 MFMA work (power management
// holding the shader clock down)?  Kernel B: one wave per CU runs a dependent chain of N fp64 FMAs; its duration is
// taken with the constant 100 MHz counter (wall_clock64).  Kernel A: all CUs issue fp64 MFMAs for a few milliseconds.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_burn(int iters, double *sink)
{
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    d4 t = a0 + a1 + a2 + a3;
    if (t[0] + t[1] + t[2] + t[3] == 123.456) sink[0] = t[0];
}
__global__ __launch_bounds__(64) void valu_chain(int n, double *sink, long long *ticks)
{
    double v = 1.0 + threadIdx.x * 1e-12;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v = __builtin_fma(v, 0.9999999, 1e-9);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (v == 123.456) sink[0] = v;
}
int main()
{
    double *sink; long long *ticks, h[256];
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ticks, 256 * 8));
    const int N = 200000;  // dependent FMAs
    auto runB = [&](const char *what) {
        hipLaunchKernelGGL(valu_chain, dim3(256), dim3(64), 0, 0, N, sink, ticks);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        s /= 256;
        printf("%-34s %8.1f us for %d dependent FMAs -> %.2f ns per FMA\n", what, s / 100.0, N, s * 10.0 / N);
    };
    runB("cold");
    runB("second run");
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(mfma_burn, dim3(512), dim3(256), 0, 0, 400000, sink);  // a few ms of dense fp64 MFMA
        hipLaunchKernelGGL(valu_chain, dim3(256), dim3(64), 0, 0, N, sink, ticks);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        s /= 256;
        printf("%-34s %8.1f us -> %.2f ns per FMA\n", "right after the MFMA burst", s / 100.0, s * 10.0 / N);
    }
    // interleaved at the sweep's granularity: 25 us of MFMA, then a short VALU kernel, many times
    double acc = 0; int cnt = 0;
    for (int rep = 0; rep < 200; ++rep) {
        hipLaunchKernelGGL(mfma_burn, dim3(512), dim3(256), 0, 0, 2000, sink);
        hipLaunchKernelGGL(valu_chain, dim3(256), dim3(64), 0, 0, 20000, sink, ticks);
        if (rep >= 100) { hipDeviceSynchronize(); hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < 256; ++i) s += h[i]; acc += s / 256; ++cnt; }
    }
    printf("%-34s %.2f ns per FMA (20000-FMA kernels between 2000-iteration MFMA kernels)\n", "interleaved", acc / cnt * 10.0 / 20000);
    return 0;
}
