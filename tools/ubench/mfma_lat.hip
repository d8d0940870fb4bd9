// micro-benchmark (tools only): issue / latency of v_mfma_f64_16x16x4_f64 and of the scalar decision chain
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lat mfma_lat.hip ; run: ./mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(64) void k(int iters, double *sink, long long *out)
{
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 1e-9, y = 1.0 - lane * 1e-9;
    d4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (d4){0, 0, 0, 0};
    long long t[12];
    long long rt0 = __builtin_amdgcn_s_memrealtime();
    // T1: independent
    t[0] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = MFMA(x, y, a[i]);
    }
    t[1] = __builtin_amdgcn_s_memtime();
    // T2: dependent chain on one accumulator
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[0] = MFMA(x, y, a[0]);
    }
    t[2] = __builtin_amdgcn_s_memtime();
    // T3: MFMA -> readlane -> short dependent VALU chain -> operand of next MFMA
    double xx = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[1] = MFMA(xx, y, a[1]);
            const double d = readlane_d(a[1][0], 5);
            xx = 1.0 + 1e-30 * d;
        }
    }
    t[3] = __builtin_amdgcn_s_memtime();
    // T4: the attractive decision chain alone (dependent), with a division
    double d = x, accum = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double dd = readlane_d(d, 5);
            const double r = 1.0 + 0.88 * (1.0 - dd);
            const double p = 1.3 * (r * r);
            const double xq = 0.88 / r;
            if (p > 1.0) accum += 1e-30;
            d = xq * 1e-30 + 0.5;
        }
    }
    t[4] = __builtin_amdgcn_s_memtime();
    // T5: independent v_fma_f64
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = f[i] * y + x;
    }
    t[5] = __builtin_amdgcn_s_memtime();
    // T6: dependent v_fma_f64
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[0] = f[0] * y + x;
    }
    t[6] = __builtin_amdgcn_s_memtime();
    // T7: MFMA with one dependent operand prepared by v_mul + cndmask (mask + scale) each
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double m = ((lane >> 4) == (i & 3)) ? a[2][i & 3] : 0.0;
            a[2] = MFMA(m * 1e-30, y, a[2]);
        }
    }
    t[7] = __builtin_amdgcn_s_memtime();
    long long rt1 = __builtin_amdgcn_s_memrealtime();
    d4 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    double fs = 0; for (int i = 0; i < 8; ++i) fs += f[i];
    if (s[0] + s[1] + s[2] + s[3] + fs + accum + d == 123.456) sink[0] = s[0];
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) out[blockIdx.x * 16 + i] = t[i];
        out[blockIdx.x * 16 + 8] = rt1 - rt0;
    }
}
int main()
{
    double *sink; long long *out;
    hipMalloc(&sink, 64); hipMalloc(&out, 2048 * 16 * 8);
    const int iters = 2000;
    for (int blocks : {1, 32, 1024, 2048}) {
        k<<<blocks, 64>>>(iters, sink, out); hipDeviceSynchronize();
        k<<<blocks, 64>>>(iters, sink, out); hipDeviceSynchronize();
        std::vector<long long> h(blocks * 16);
        hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost);
        const char *names[] = {"indep MFMA", "dep MFMA", "MFMA->readlane->VALU->MFMA", "decision chain (div)", "indep fma64", "dep fma64", "mask+scale+dep MFMA"};
        const double n = iters * 8.0;
        const double cyc_total = (double)(h[7] - h[0]), real = (double)h[8] * 10e-9;  // 100 MHz
        printf("blocks=%d  clock ~ %.2f GHz\n", blocks, cyc_total / real / 1e9);
        for (int i = 0; i < 7; ++i) printf("  %-32s %.1f cycles each\n", names[i], (h[i + 1] - h[i]) / n);
    }
    return 0;
}
