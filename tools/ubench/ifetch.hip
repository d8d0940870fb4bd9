// ifetch.hip - how fast does a LONE wave issue straight-line code that is executed once (cold in the instruction cache),
// against the same instructions in a loop that stays in the cache?  One wave per workgroup, one workgroup per CU.
// Build / run: hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=4000000 -o ifetch ifetch.hip && ./ifetch
// Prints shader-clock cycles per instruction for: fp64 FMA (8-byte VOP3 encoding), independent chains of 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FMA(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[(i) & 7]) : "v"(x), "v"(y))
constexpr int N_COLD = 16384;   // 128 KB of code: twice the instruction cache
constexpr int N_BODY = 64, N_ITER = N_COLD / N_BODY;

__global__ __launch_bounds__(64) void cold_kernel(double *out, long long *cyc, double x, double y)
{
    double a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_COLD; ++i) FMA(i);
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(64) void hot_kernel(double *out, long long *cyc, double x, double y)
{
    double a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < N_BODY; ++i) FMA(i);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const int blocks = 256;
    double *out; long long *cyc;
    hipMalloc(&out, blocks * 64 * sizeof(double));
    hipMalloc(&cyc, blocks * sizeof(long long));
    std::vector<long long> h(blocks);
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            if (which == 0) hipLaunchKernelGGL(cold_kernel, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
            else hipLaunchKernelGGL(hot_kernel, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
            long long mn = h[0], mx = h[0]; double av = 0;
            for (auto v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; av += v; }
            printf("%s: %d v_fma_f64 per wave, cycles per instruction min %.2f avg %.2f max %.2f\n", which ? "loop (64 per iteration)" : "straight line (executed once)",
                   N_COLD, (double)mn / N_COLD, av / blocks / N_COLD, (double)mx / N_COLD);
        }
    }
    return 0;
}
