// micro-benchmark (tools only): issue cost of f64 VALU work on one CU at 1, 2 and 4 waves per SIMD
// build: hipcc --offload-arch=gfx950 -O3 -o valu_f64 valu_f64.hip ; run: ./valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, double *sink, long long *out)
{
    double a[16];
    const double x = 1.0 + threadIdx.x * 1e-12, y = 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i + y;
    int m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = threadIdx.x + i;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 16 independent f64 FMA
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], x, y);
        } else if (MODE == 1) {  // 8 f64 FMA + 8 int ops
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = __builtin_fma(a[i], x, y);
                m[i] = (m[i] * 3) ^ it;
            }
        } else if (MODE == 2) {  // 16 int ops (mul_lo + xor)
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = (m[i] + 3) ^ it;
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = (m[i] + 5) ^ it;
        } else if (MODE == 4) {  // 16 DEPENDENT f64 fma
#pragma unroll
            for (int i = 0; i < 16; ++i) a[0] = __builtin_fma(a[0], x, y);
        } else if (MODE == 5) {  // 16 dependent v_cndmask pairs (select chain on f64)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[0] = (m[0] == it + i) ? a[1] : a[0] + 0.0 * a[2];
        } else if (MODE == 6) {  // 8 dependent (dpp pair + f64 max): one wave-reduction level each
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lo = __double2loint(a[0]), hi = __double2hiint(a[0]);
                const double o = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x118, 0xf, 0xf, false),
                                                  __builtin_amdgcn_update_dpp(lo, lo, 0x118, 0xf, 0xf, false));
                a[0] = fmax(a[0], o) + y;
            }
        } else {  // 16 f64 mul (not fma)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = a[i] * x;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    int ms = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) ms += m[i];
    sink[threadIdx.x] = s + ms;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
int main()
{
    double *sink;
    long long *out, h[16];
    (void)0;
    hipMalloc(&sink, 1024 * 8);
    hipMalloc(&out, 16 * 8);
    const int iters = 2000;
    const char *names[7] = {"16 indep f64 fma", "8 f64 fma + 8x(int mul, xor)", "16x(int add, xor)", "16 indep f64 mul", "16 dependent f64 fma", "16 dependent f64 selects(+add)", "8 dependent dpp+max+add levels"};
    for (int mode = 0; mode < 7; ++mode)
        for (int threads = 256; threads <= 512; threads *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, iters, sink, out);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, out, 16 * 8, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
            printf("%-32s waves/SIMD %d: memtime ticks per loop iteration: wave 0 %.2f, slowest wave %.2f\n", names[mode],
                   threads / 256, (double)h[0] / iters, (double)mx / iters);
        }
    return 0;
}
