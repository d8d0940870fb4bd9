// qr.hip — column-pivoted Householder QR for the UDT decomposition
// (udt_AVX_pivot!, src/linalg/UDT.jl:192-306) and the pivot-aware triangular
// right-solve rdivp! (src/linalg/general.jl:138-166).
//
// One workgroup (1024 threads, 16 waves) factors one n x n matrix.  The pivot
// rule is the reference's: at every step the norms of ALL trailing columns over
// rows j..n are taken from the fully updated matrix (indmaxcolumn, UDT.jl:151-168,
// no LAPACK-style down-dating) and the first maximum wins.  The norm pass of step
// j+1 is fused into the reflector application of step j (the updated column is in
// registers anyway), so every trailing element is read once and written once per
// step.  Column norms and reflector dot products are wavefront shuffle reductions.
#include "kernels.h"

namespace dqmc {

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

constexpr int QR_THREADS = 1024;
constexpr int QR_WAVES = QR_THREADS / 64;

// QMAX = rows per lane (n <= 64*QMAX); QR_UC = trailing columns a wave keeps in flight
template <int QMAX, int QR_UC>
__global__ __launch_bounds__(QR_THREADS) void qr_pivot_kernel(int n, double *__restrict__ Aall, long strideA,
                                                             double *__restrict__ tauall,
                                                             int *__restrict__ pivall)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *v = sm;             // current reflector, v[j] = 1, v[r<j] = 0
    double *norms = sm + 1024;  // squared norms of trailing columns over rows >= j
    __shared__ int s_jm;
    __shared__ double s_max;

    const int unit = blockIdx.x;
    double *__restrict__ A = Aall + (long)unit * strideA;
    double *__restrict__ tau = tauall + (long)unit * n;
    int *__restrict__ piv = pivall + (long)unit * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < n; i += QR_THREADS) piv[i] = i;
    for (int k = wave; k < n; k += QR_WAVES) {
        double s = 0.0;
        for (int r = lane; r < n; r += 64) {
            const double a = A[r + (long)n * k];
            s += a * a;
        }
        s = wave_sum(s);
        if (lane == 0) norms[k] = s;
    }
    __syncthreads();

    for (int j = 0; j < n; ++j) {
        // ---- pivot: first maximum of the trailing column norms (UDT.jl:151-168)
        if (wave == 0) {
            double best = -1.0;
            int bi = 0x7fffffff;
            for (int k = j + lane; k < n; k += 64) {
                const double val = norms[k];
                if (val > best) { best = val; bi = k; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) {
                s_jm = (bi < n) ? bi : j;
                s_max = (bi < n) ? best : 0.0;
            }
        }
        __syncthreads();
        const int jm = s_jm;
        const double maxval = s_max;

        // ---- swap columns j <-> jm (UDT.jl:219-231) and build the reflector (UDT.jl:133-148)
        const int r = tid;
        const double xi1 = A[j + (long)n * jm];
        double a = 0.0, b = 0.0;
        if (r < n) {
            a = A[r + (long)n * jm];
            b = A[r + (long)n * j];
        }
        __syncthreads();
        double tj = 0.0, nu = 0.0, xi = 1.0;
        if (maxval != 0.0) {
            nu = copysign(sqrt(maxval), xi1);
            xi = xi1 + nu;
            tj = xi / nu;
        }
        if (r < n) {
            double newj = a, vr = (r == j) ? 1.0 : 0.0;
            if (maxval != 0.0) {
                if (r == j) newj = -nu;
                else if (r > j) { newj = a / xi; vr = newj; }
            }
            A[r + (long)n * j] = newj;
            if (jm != j) A[r + (long)n * jm] = b;
            v[r] = vr;
        }
        if (tid == 0) {
            tau[j] = tj;
            const int t = piv[j];
            piv[j] = piv[jm];
            piv[jm] = t;
        }
        __syncthreads();

        // ---- apply H_j to the trailing columns (reflectorApply!, UDT.jl:32-50) and
        //      take the norms needed by step j+1 from the updated registers
        const int qn = (n - j + 63) >> 6;
        for (int k0 = j + 1 + wave * QR_UC; k0 < n; k0 += QR_WAVES * QR_UC) {
            double x[QR_UC][QMAX];
            double dot[QR_UC];
#pragma unroll
            for (int c = 0; c < QR_UC; ++c) {
                dot[c] = 0.0;
                const int k = k0 + c;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    const int rr = j + lane + 64 * q;
                    x[c][q] = (q < qn && k < n && rr < n) ? A[rr + (long)n * k] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int rr = j + lane + 64 * q;
                const double vv = (q < qn && rr < n) ? v[rr] : 0.0;
#pragma unroll
                for (int c = 0; c < QR_UC; ++c) dot[c] += vv * x[c][q];
            }
#pragma unroll
            for (int c = 0; c < QR_UC; ++c) dot[c] = wave_sum(dot[c]) * tj;
            double nrm[QR_UC];
#pragma unroll
            for (int c = 0; c < QR_UC; ++c) nrm[c] = 0.0;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int rr = j + lane + 64 * q;
                const double vv = (q < qn && rr < n) ? v[rr] : 0.0;
#pragma unroll
                for (int c = 0; c < QR_UC; ++c) {
                    const double y = x[c][q] - vv * dot[c];
                    x[c][q] = y;
                    if (rr > j) nrm[c] += y * y;
                }
            }
#pragma unroll
            for (int c = 0; c < QR_UC; ++c) {
                const int k = k0 + c;
                if (k < n) {
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        const int rr = j + lane + 64 * q;
                        if (q < qn && rr < n) A[rr + (long)n * k] = x[c][q];
                    }
                }
                const double s = wave_sum(nrm[c]);
                if (lane == 0 && k < n) norms[k] = s;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_qr_pivot(int n, int n_units, double *A, long strideA, double *tau, int *pivot, hipStream_t s)
{
    if (n > 1024) return hipErrorInvalidValue;
    const size_t lds = 2 * 1024 * sizeof(double);
    dim3 grid(n_units), block(QR_THREADS);
#define QR_LAUNCH(Q, UC) hipLaunchKernelGGL((qr_pivot_kernel<Q, UC>), grid, block, lds, s, n, A, strideA, tau, pivot)
    if (n <= 64) QR_LAUNCH(1, 4);
    else if (n <= 128) QR_LAUNCH(2, 4);
    else if (n <= 256) QR_LAUNCH(4, 4);
    else if (n <= 576) QR_LAUNCH(9, 2);
    else QR_LAUNCH(16, 1);
#undef QR_LAUNCH
    return hipGetLastError();
}

// D, V and T from the factored matrix (UDT.jl:268-306).  One workgroup per unit.
__global__ __launch_bounds__(1024) void udt_finish_kernel(int n, double *__restrict__ Aall, long strideA,
                                                         const int *__restrict__ pivall,
                                                         double *__restrict__ Dall, long strideD,
                                                         double *__restrict__ Vall, long strideV,
                                                         double *__restrict__ Tall, long strideT, int apply_pivot)
{
    extern __shared__ __attribute__((aligned(16))) double dinv[];  // 1/D
    const int unit = blockIdx.x;
    double *__restrict__ A = Aall + (long)unit * strideA;
    double *__restrict__ D = Dall + (long)unit * strideD;
    const int *__restrict__ piv = pivall + (long)unit * n;
    double *__restrict__ V = Vall ? Vall + (long)unit * strideV : nullptr;
    double *__restrict__ T = Tall ? Tall + (long)unit * strideT : nullptr;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = fabs(A[i + (long)n * i]);
        D[i] = d;
        dinv[i] = 1.0 / d;
    }
    __syncthreads();
    const int rpt = (n + 63) / 64;  // rows handled per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wave; c < n; c += nw) {
        const int pc = apply_pivot ? piv[c] : c;
        for (int q = 0; q < rpt; ++q) {
            const int r = lane + 64 * q;
            if (r >= n) break;
            const double a = A[r + (long)n * c];
            if (V) V[r + (long)n * c] = r > c ? a : (r == c ? 1.0 : 0.0);
            if (apply_pivot) T[r + (long)n * pc] = r <= c ? dinv[r] * a : 0.0;
            else if (r <= c) A[r + (long)n * c] = dinv[r] * a;
        }
    }
}

hipError_t launch_udt_finish(int n, int n_units, double *A, long strideA, const int *pivot, double *D,
                             long strideD, double *V, long strideV, double *Tout, long strideT,
                             int apply_pivot, hipStream_t s)
{
    hipLaunchKernelGGL(udt_finish_kernel, dim3(n_units), dim3(1024), n * sizeof(double), s, n, A, strideA, pivot,
                       D, strideD, V, strideV, Tout, strideT, apply_pivot);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// X = A[:, pivot] * triu(T)^-1.  Rows are independent: one workgroup owns a slab of
// RS rows, keeps it in LDS as Xs[col][row] and walks the columns; the k-sum of
// column j is split over KS = 256/RS thread groups and combined through LDS.
template <int RS>
__global__ __launch_bounds__(256) void trsm_kernel(int n, const double *__restrict__ Aall, long sA,
                                                  const double *__restrict__ Tall, long sT,
                                                  const int *__restrict__ pivall,
                                                  const double *__restrict__ dmulall, long sV,
                                                  double *__restrict__ Oall, long sO, int slabs)
{
    constexpr int KS = 256 / RS;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm;                          // [n][RS]
    double *Tc = Xs + (size_t)n * RS;         // [n rounded up]
    double *part = Tc + ((n + 63) & ~63);     // [KS][RS]
    const int unit = blockIdx.x / slabs, row0 = (blockIdx.x % slabs) * RS;
    const double *__restrict__ A = Aall + (long)unit * sA;
    const double *__restrict__ T = Tall + (long)unit * sT;
    const int *__restrict__ piv = pivall ? pivall + (long)unit * n : nullptr;
    const double *__restrict__ dmul = dmulall ? dmulall + (long)unit * sV : nullptr;
    double *__restrict__ O = Oall + (long)unit * sO;
    const int tid = threadIdx.x;

    for (int idx = tid; idx < n * RS; idx += 256) {
        const int j = idx / RS, r = idx - j * RS, row = row0 + r;
        const int pj = piv ? piv[j] : j;
        Xs[idx] = row < n ? A[row + (long)n * pj] : 0.0;
    }
    const int r = tid % RS, q = tid / RS;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        for (int k = tid; k <= j; k += 256) Tc[k] = T[k + (long)n * j];
        __syncthreads();
        double s = 0.0;
        for (int k = q; k < j; k += KS) s += Xs[k * RS + r] * Tc[k];
        part[q * RS + r] = s;
        __syncthreads();
        if (q == 0) {
            double tot = 0.0;
#pragma unroll
            for (int qq = 0; qq < KS; ++qq) tot += part[qq * RS + r];
            double x = Xs[j * RS + r] - tot;
            x = dmul ? x * dmul[j] : x / Tc[j];
            Xs[j * RS + r] = x;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * RS; idx += 256) {
        const int j = idx / RS, rr = idx - j * RS, row = row0 + rr;
        if (row < n) O[row + (long)n * j] = Xs[idx];
    }
}

hipError_t launch_trsm_right_upper(int n, int n_units, const double *A, long sA, const double *T, long sT,
                                   const int *pivot, const double *dmul, long sV, double *Out, long sO,
                                   hipStream_t s)
{
    // slab height: largest of 64/32/16 whose LDS image fits
    const size_t budget = 150 * 1024;
    auto need = [&](int rs) { return ((size_t)n * rs + ((n + 63) & ~63) + 256) * sizeof(double); };
    int rs = 64;
    while (rs > 16 && need(rs) > budget) rs >>= 1;
    if (need(rs) > budget) return hipErrorInvalidValue;
    const int slabs = (n + rs - 1) / rs;
    dim3 grid(n_units * slabs), block(256);
    const size_t lds = need(rs);
#define TR_LAUNCH(RS)                                                                                         \
    do {                                                                                                      \
        (void)hipFuncSetAttribute((const void *)trsm_kernel<RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((trsm_kernel<RS>), grid, block, lds, s, n, A, sA, T, sT, pivot, dmul, sV, Out, sO, slabs); \
    } while (0)
    if (rs == 64) TR_LAUNCH(64);
    else if (rs == 32) TR_LAUNCH(32);
    else TR_LAUNCH(16);
#undef TR_LAUNCH
    return hipGetLastError();
}

}  // namespace dqmc
