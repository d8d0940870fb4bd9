// rcp_acc.hip - accuracy of v_rcp_f64 / v_rsq_f64 (bits), raw and after one / two Newton steps
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // x spread over [0.25, 4) and over many exponents
    const double x = (0.25 + 3.75 * (i + 0.5) / n) * ((i % 7 == 0) ? 1e-50 : (i % 11 == 0 ? 1e80 : 1.0)) * ((i & 1) ? -1.0 : 1.0);
    double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    const double t = 1.0 / x;
    out[3 * i] = fabs(r0 - t) / fabs(t);
    out[3 * i + 1] = fabs(r1 - t) / fabs(t);
    out[3 * i + 2] = fabs(r2 - t) / fabs(t);
}
int main()
{
    const int n = 1 << 22;
    double *d; hipMalloc(&d, 3 * n * sizeof(double));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double *h = (double *)malloc(3 * n * sizeof(double));
    hipMemcpy(h, d, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) m[j] = fmax(m[j], h[3 * i + j]);
    printf("v_rcp_f64 max rel err: raw %.3e (2^%.1f), 1 Newton %.3e (2^%.1f), 2 Newton %.3e\n", m[0], log2(m[0]), m[1], log2(m[1]), m[2]);
    return 0;
}
