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
#include <cstdlib>

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
                                                             int *__restrict__ pivall,
                                                             const double *__restrict__ srcall, long strideSrc,
                                                             int *guard, int guard_val)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // fallback use: runs only if the cooperative kernel of this launch epoch timed out, on a fresh copy of the input
    if (guard && guard[0] != guard_val) return;
    if (srcall) {
        const double *__restrict__ src = srcall + (long)blockIdx.x * strideSrc;
        double *__restrict__ dst = Aall + (long)blockIdx.x * strideA;
        for (long i = threadIdx.x; i < (long)n * n; i += QR_THREADS) dst[i] = src[i];
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&guard[1], 1);
        __syncthreads();
    }
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

// ---------------------------------------------------------------------------
// Panel form for n > 256 (config 5: n = 576), one workgroup per matrix.  The streaming kernel above moves the whole
// trailing matrix through the workgroup twice per step (read + write): at n = 576 with 256 matrices that is 390 GB per
// batch, i.e. the kernel sits on the HBM roofline of the unblocked algorithm (49.5 ms per batch, 5.3 TB/s).  This one
// keeps the reference's pivot rule but reorganises the arithmetic the way LAPACK's dlaqps does, so that a step READS the
// trailing matrix once and writes nothing but one row:
//   * panels of QP_NB columns; inside a panel the reflectors are NOT applied to the trailing matrix.  Step j = j0 + k:
//     pivot = first maximum of the trailing column norms; the pivot column alone is brought up to date
//     (a -= V[:, 0:k] F[j, 0:k]'), its norm is taken from scratch (as the reference takes it, UDT.jl:151-168) and the
//     reflector built (UDT.jl:133-148); F[:, k] = tau A' v over the not yet updated trailing matrix - the one pass
//     over memory -, corrected by the earlier reflectors of the panel; row j of the trailing matrix is updated
//     (it is row j of R);
//   * the norms of the other columns are DOWN-DATED with that row (|a_c|^2 -= r_jc^2) instead of recomputed - the one
//     departure from the reference's arithmetic (it recomputes every norm from the updated matrix at every step).  A
//     column whose norm has lost four digits since it was last computed exactly (|a_c|^2 <= 1e-4 of the reference value;
//     LAPACK tolerates sqrt(eps)) is recomputed on the spot from the virtually updated column, so the norms that decide
//     a pivot agree with the reference's to ~1e-9 relative and the pivot can differ only on such near-ties;
//   * at the end of a panel the trailing matrix gets the rank-QP_NB update A -= V F' (one read + one write).
// Traffic: (1 + 2 / QP_NB) passes per step instead of 2.  Arithmetic is negligible (a CU needs ~2 flops per cycle to keep up).
// Output format as qr_pivot_kernel: R on / above the diagonal, Householder vectors below (unit diagonal implied).
constexpr int QP_NB = 16;
constexpr int QP_UC = 8;   // columns a wave keeps in flight in the pass over the trailing matrix: 72 requests per lane - the
                           // loop does not overlap its iterations, so a group's requests are all the memory parallelism there is
                           // (2 columns: 48 ms per batch at n = 576, latency-bound at 3 TB/s)
constexpr int QP_UC2 = 4;  // columns per group in the rank-QP_NB update
constexpr int QP_THREADS = 512, QP_WAVES = QP_THREADS / 64;  // 2 waves per SIMD: 256 VGPRs per lane (at 1024 threads the
                                                             // 128-register budget spilled and the kernel ran 2.4 x slower)
constexpr int QP_RPT = 2;  // rows per thread in the one-element-per-row phases (n <= 1024)
// barrier that hands over LDS data only (the global stores of the step stay in flight; the one full barrier of a step
// sits behind the row update, before the next step touches what this one stored)
#define QP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int QMAX>
__global__ __launch_bounds__(QP_THREADS) void qr_panel_kernel(int n, double *__restrict__ Aall, long strideA,
                                                             double *__restrict__ tauall, int *__restrict__ pivall,
                                                             double thr)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *F = sm;                  // [QP_NB][n]: F[kk * n + c]
    double *vn1 = F + QP_NB * n;     // squared partial norms (down-dated)
    double *vn2 = vn1 + n;           // their values when last computed exactly
    double *v = vn2 + n;             // current reflector: v[j] = 1, v[r < j] = 0
    double *rowj = v + n;            // row j of the trailing matrix as the pass over it found it
    double *aux = rowj + n;          // [2 QP_NB]: aux of the F update, row j of the panel's reflectors
    double *red = aux + 2 * QP_NB;   // [QP_WAVES + 2]
    int *pvw = reinterpret_cast<int *>(red + QP_WAVES + 2);   // [QP_WAVES] per-wave pivot candidates
    int *todo = pvw + QP_WAVES;      // columns whose norm must be recomputed, [0] = count

    const int unit = blockIdx.x;
    double *__restrict__ A = Aall + (long)unit * strideA;
    double *__restrict__ tau = tauall + (long)unit * n;
    int *__restrict__ piv = pivall + (long)unit * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < n; i += QP_THREADS) piv[i] = i;
    for (int c = wave; c < n; c += QP_WAVES) {
        double s = 0.0;
        for (int r = lane; r < n; r += 64) {
            const double a = A[r + (long)n * c];
            s += a * a;
        }
        s = wave_sum(s);
        if (lane == 0) { vn1[c] = s; vn2[c] = s; }
    }
    if (tid == 0) todo[0] = 0;
    __syncthreads();

    for (int j0 = 0; j0 < n; j0 += QP_NB) {
        const int kb = min(QP_NB, n - j0);
        for (int i = tid; i < kb * n; i += QP_THREADS) F[i] = 0.0;
        __syncthreads();
        for (int k = 0; k < kb; ++k) {
            const int j = j0 + k;
            // ---- pivot: first maximum of the trailing column norms (UDT.jl:151-168); every wave scans a stripe, the
            // eight partial results meet in LDS and every thread finishes the search for itself
            {
                double best = -1.0;
                int bi = 0x7fffffff;
                for (int c = j + tid; c < n; c += QP_THREADS) {
                    const double val = vn1[c];
                    if (val > best) { best = val; bi = c; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                if (lane == 0) { red[wave] = best; pvw[wave] = bi; }
            }
            QP_LDS_BARRIER();
            int jm;
            {
                double best = red[0];
                int bi = pvw[0];
#pragma unroll
                for (int w2 = 1; w2 < QP_WAVES; ++w2) {
                    const double ov = red[w2];
                    const int oi = pvw[w2];
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                jm = (bi < n) ? bi : j;
            }
            // ---- the pivot column (now at position jm) comes into registers, is brought up to date with the reflectors
            // of this panel (rows >= j) and goes to position j; the column that was there goes to jm whole - the swap of
            // UDT.jl:219-231 without a pass of its own.  (All loops over the panel run over the compile-time QP_NB with
            // clamped addresses and zero coefficients: a run-time trip count, or a predicate on a load, turns sixteen
            // independent requests into sixteen round trips.)
            double a[QP_RPT], bold[QP_RPT];
            double sq = 0.0;
            double fj[QP_NB];
#pragma unroll
            for (int kk = 0; kk < QP_NB; ++kk) fj[kk] = kk < k ? F[kk * n + jm] : 0.0;
#pragma unroll
            for (int e = 0; e < QP_RPT; ++e) {
                const int r = tid + QP_THREADS * e;
                int rc = min(r, n - 1);
                // 32-bit element offsets from the matrix's scalar base, recomputed in every step (rc is made opaque): the
                // 34 addresses of this block are invariant inside a panel, the compiler hoisted them out of the step loop,
                // spilled them, and every request then waited for the reload of its own address - 15 trips to the L2
                // one after the other in every step
                asm volatile("" : "+v"(rc));
                a[e] = A[(unsigned)(rc + n * jm)];
                bold[e] = A[(unsigned)(rc + n * j)];
                double vp[QP_NB];
#pragma unroll
                for (int kk = 0; kk < QP_NB; ++kk) vp[kk] = A[(unsigned)(rc + n * (j0 + min(kk, kb - 1)))];
#pragma unroll
                for (int kk = 0; kk < QP_NB; ++kk) a[e] -= (r >= j ? fj[kk] : 0.0) * vp[kk];
                sq += (r < n && r >= j) ? a[e] * a[e] : 0.0;
            }
            QP_LDS_BARRIER();  // (everybody has read red[] / pvw[] / F[., jm])
#pragma unroll
            for (int e = 0; e < QP_RPT; ++e)
                if (tid + QP_THREADS * e == j) red[QP_WAVES] = a[e];
            // its squared norm over rows >= j, from scratch, and element j (block reduction in fixed order)
            sq = wave_sum(sq);
            if (lane == 0) red[wave] = sq;
            if (jm != j) {
                if (tid < kb) {
                    const double fa = F[tid * n + jm], fb = F[tid * n + j];
                    F[tid * n + j] = fa;
                    F[tid * n + jm] = fb;
                }
                if (tid == 0) {
                    vn1[jm] = vn1[j]; vn2[jm] = vn2[j];
                    const int t = piv[j]; piv[j] = piv[jm]; piv[jm] = t;
                }
#pragma unroll
                for (int e = 0; e < QP_RPT; ++e) {
                    const int r = tid + QP_THREADS * e;
                    if (r < n) A[r + (long)n * jm] = bold[e];
                }
            }
            // row j of the earlier reflectors of the panel (needed for the row update below)
            if (tid < QP_NB) aux[QP_NB + tid] = tid < k ? A[j + (long)n * (j0 + tid)] : 0.0;
            __syncthreads();
            double maxval = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < QP_WAVES; ++w2) maxval += red[w2];
            const double xi1 = red[QP_WAVES];
            // ---- reflector (UDT.jl:133-148)
            double tj = 0.0, nu = 0.0, xi = 1.0;
            if (maxval != 0.0) {
                nu = copysign(sqrt(maxval), xi1);
                xi = xi1 + nu;
                tj = xi / nu;
            }
#pragma unroll
            for (int e = 0; e < QP_RPT; ++e) {
                const int r = tid + QP_THREADS * e;
                if (r < n) {
                    double vr = (r == j) ? 1.0 : 0.0, newj = a[e];
                    if (r >= j && maxval != 0.0) {
                        if (r == j) newj = -nu;
                        else { newj = a[e] / xi; vr = newj; }
                    }
                    if (r >= j || jm != j) A[r + (long)n * j] = newj;  // (rows < j: the R entries the column brought along)
                    v[r] = vr;
                }
            }
            if (tid == 0) tau[j] = tj;
            QP_LDS_BARRIER();  // (v is in LDS; the column-j stores of this step are not read before the next full barrier)
            // ---- F[k][c] = tau A[j:, c]' v for the trailing columns c > j: the ONE pass over the trailing matrix.
            // Every request of a column group goes out before the first product (clamped addresses, no predicates).
            double vq[QMAX];
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int rr = j + lane + 64 * q;
                vq[q] = rr < n ? v[min(rr, n - 1)] : 0.0;
            }
            // (the pass starts at the panel's first column: for c < j the same product is aux[c - j0] = -tau V[:, c]' v,
            // and lane 0 of the first row block holds A[j, c], which the row update below needs)
            for (int c0 = j0 + wave * QP_UC; c0 < n; c0 += QP_WAVES * QP_UC) {
                double xa[QP_UC][QMAX];
#pragma unroll
                for (int u = 0; u < QP_UC; ++u) {
                    const double *__restrict__ col = A + (long)n * min(c0 + u, n - 1);
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) xa[u][q] = col[min(j + lane + 64 * q, n - 1)];
                }
#pragma unroll
                for (int u = 0; u < QP_UC; ++u) {
                    double d = 0.0;
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) d += vq[q] * xa[u][q];
                    d = wave_sum(d);
                    const int c = c0 + u;
                    if (lane == 0 && c < n) {
                        if (c > j) { F[k * n + c] = tj * d; rowj[c] = xa[u][0]; }
                        else if (c < j) aux[c - j0] = -tj * d;
                    }
                }
            }
            QP_LDS_BARRIER();
            // ---- F[:, k] += F[:, 0:k] aux; row j of the trailing matrix (= row j of R); norm down-dating
            for (int c = j + 1 + tid; c < n; c += QP_THREADS) {
                double f = F[k * n + c], s2 = 0.0;
                const double ajc = rowj[c];
#pragma unroll
                for (int kk = 0; kk < QP_NB; ++kk) {
                    const double fk = kk < k ? F[kk * n + c] : 0.0;
                    f += fk * aux[kk];
                    s2 += fk * aux[QP_NB + kk];
                }
                F[k * n + c] = f;
                const double rjc = ajc - f - s2;  // v[j] = 1
                A[j + (long)n * c] = rjc;
                const double t = vn1[c] - rjc * rjc;
                if (t <= thr * vn2[c]) todo[1 + atomicAdd(&todo[0], 1)] = c;  // cancellation: take it from scratch
                else vn1[c] = t;
            }
            __syncthreads();
            // ---- exact norms of the flagged columns, from the virtually updated column
            const int ntodo = todo[0];
            for (int i = wave; i < ntodo; i += QP_WAVES) {
                const int c = todo[1 + i];
                double sacc = 0.0;
                for (int rr = j + 1 + lane; rr < n; rr += 64) {
                    double x = A[rr + (long)n * c];
                    for (int kk = 0; kk <= k; ++kk) x -= A[rr + (long)n * (j0 + kk)] * F[kk * n + c];
                    sacc += x * x;
                }
                sacc = wave_sum(sacc);
                if (lane == 0) { vn1[c] = sacc; vn2[c] = sacc; }
            }
            if (ntodo) {
                __syncthreads();
                if (tid == 0) todo[0] = 0;
                __syncthreads();
            }
        }
        // ---- rank-kb update of the trailing matrix below / right of the panel: A[r, c] -= V[r, :] F[:, c]
        const int jn = j0 + kb;
        if (jn < n) {
            for (int c0 = jn + wave * QP_UC2; c0 < n; c0 += QP_WAVES * QP_UC2) {
                // (1024 threads = 4 waves per SIMD = 128 VGPRs per lane: the panel's F entries come from LDS as
                // broadcast reads instead of sitting in registers)
                double x[QP_UC2][QMAX];
#pragma unroll
                for (int u = 0; u < QP_UC2; ++u) {
                    const double *__restrict__ col = A + (long)n * min(c0 + u, n - 1);
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) x[u][q] = col[min(jn + lane + 64 * q, n - 1)];
                }
#pragma unroll 4
                for (int kk = 0; kk < kb; ++kk) {
                    const double *__restrict__ vc = A + (long)n * (j0 + kk);
                    double vk[QMAX];
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) vk[q] = vc[min(jn + lane + 64 * q, n - 1)];
#pragma unroll
                    for (int u = 0; u < QP_UC2; ++u) {
                        const double fk = F[kk * n + min(c0 + u, n - 1)];
#pragma unroll
                        for (int q = 0; q < QMAX; ++q) x[u][q] -= vk[q] * fk;
                    }
                }
#pragma unroll
                for (int u = 0; u < QP_UC2; ++u)
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        const int rr = jn + lane + 64 * q;
                        if (rr < n && c0 + u < n) A[rr + (long)n * (c0 + u)] = x[u][q];
                    }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// On-chip variant for 128 < n <= 256: one workgroup of 512 threads (8 waves) keeps the matrix
// next to the ALUs for the whole factorisation, so a step costs FMAs and LDS broadcasts instead
// of an L2 round trip per trailing element.
//
//   lane l : row group rg = l & 7, column group cg = l >> 3;   wave w = 0..7
//   rows       r = rg + 8 k               k = 0..31   (32 rows per thread)
//   positions  p = 64 s + 8 w + cg        s = 0..3    (4 columns per thread)
// Storage class of a position follows its life time (position p is dead after step p) and is
// uniform per slot:   slot 0  [0,64)    stays in place in global memory / L2 (dead after 64 steps)
//                     slot 1  [64,128)  LDS, padded column stride 264
//                     slot 2,3 [128,256) registers, 2 x 32 doubles per thread
// Liveness of a slot is wave-uniform (8 w + 7 + 64 s > j), so finished columns cost nothing.
// Dot products of a column are reduced over its 8 row-group lanes with DPP (quad_perm,
// row_half_mirror); norm partials go to LDS and are summed in fixed order by the pivot search.
// A column swap j <-> jm moves 2 x 256 doubles through LDS between the 8 owner lanes of each
// column.  The pivot rule is the reference's (norms from the updated matrix, first maximum).
constexpr int QT_THREADS = 512;
#ifndef QT_GB
#define QT_GB 4  // slot-0 requests in flight per lane; must divide 4 (a region's first row block is 4 KB0).  The loop over
                 // the groups stays rolled: unrolled, the register allocator spills ten times as much
#endif
static_assert(4 % QT_GB == 0, "QT_GB must divide 4: groups start at row block 4 * KB0");
constexpr int QT_LSTRIDE = 264;  // LDS column stride in doubles (bank-conflict-free for 4 column groups)
constexpr int QT_VS = 34;        // stride of the per-row-group reflector slices (conflict-free 16-byte broadcasts)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 8 lanes {8g .. 8g+7}; every lane receives the total
__device__ __forceinline__ double sum8(double x)
{
    x += dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f64<0x141>(x);  // row_half_mirror
    return x;
}

// The reflector is kept permuted in LDS, vq[k] = v[rg + 8k] for this thread's row group, so that a
// thread fetches two of its rows per 16-byte broadcast read.
// Register-resident column: dot, update, norm partial (static indices, rows >= 32 KB0 only).
// Only the first 4 k of a region can contain finished rows (r <= j): they alone need the norm mask.
template <int KB0>
__device__ __forceinline__ void qt_reg_col(double (&x)[32], const double *vq, double *nrmp, int rg, int p, int j,
                                           double tj)
{
    double d = 0.0;
#pragma unroll
    for (int k = 4 * KB0; k < 32; k += 2) {
        const double2 v2 = *reinterpret_cast<const double2 *>(vq + k);
        d += v2.x * x[k];
        d += v2.y * x[k + 1];
    }
    const double wv = (p > j) ? sum8(d) * tj : 0.0;  // finished columns are left untouched
    double nr = 0.0;
#pragma unroll
    for (int k = 4 * KB0; k < 32; k += 2) {
        const double2 v2 = *reinterpret_cast<const double2 *>(vq + k);
        const double y0 = x[k] - v2.x * wv, y1 = x[k + 1] - v2.y * wv;
        x[k] = y0;
        x[k + 1] = y1;
        if (k < 4 * KB0 + 4) {
            nr += (rg + 8 * k > j ? 1.0 : 0.0) * (y0 * y0);
            nr += (rg + 8 * (k + 1) > j ? 1.0 : 0.0) * (y1 * y1);
        } else {
            nr += y0 * y0;
            nr += y1 * y1;
        }
    }
    nrmp[rg * 256 + p] = nr;
}

// KB0 = j / 32: first row block (4 k = 32 rows) that still has unfinished rows
// FULLN: n == 256, known at compile time: every `row < n` folds away.  With a run-time n each of the 32 global loads of the
// memory-resident slot sat in a branch of its own with a full wait behind it (32 trips to the L2 per pass, three passes
// per step during the first 64 steps: ~17 us per step against ~6 us for the later ones).
template <int KB0, bool FULLN>
__device__ __forceinline__ void qt_step(int n_rt, int j, double *__restrict__ A, double *__restrict__ tau,
                                        double *Lc, double *colP, double *colJ, double *vbuf, double *nrmp,
                                        double *dummy, int *pv, double *cand_v, int *cand_i, double (&x2)[32],
                                        double (&x3)[32])
{
    const int n = FULLN ? 256 : n_rt;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = lane & 7, cg = lane >> 3;
    const int pb = 8 * w + cg;                          // position of slot s is 64 s + pb
    double *g0 = A + (long)n * pb;                      // slot 0: in place in global memory
    double *l1 = Lc + pb * QT_LSTRIDE;                  // slot 1: LDS

    // ---- A. pivot search: first maximum of the trailing column norms (UDT.jl:151-168).
    // Wave w scans positions 32 w .. 32 w + 31 (norm partials summed in fixed order), the eight
    // wave candidates meet in LDS and every thread finishes the search redundantly.
    {
        double best = -1.0;
        int bi = 0x7fffffff;
        const int p = 32 * w + (lane & 31);
        if (lane < 32 && p >= j && p < n) {
            double val = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) val += nrmp[g * 256 + p];
            best = val;
            bi = p;
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            cand_v[w] = best;
            cand_i[w] = bi;
        }
    }
    __syncthreads();
    int jm = 0x7fffffff;
    double maxval = -1.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double ov = cand_v[q];
        const int oi = cand_i[q];
        if (ov > maxval || (ov == maxval && oi < jm)) { maxval = ov; jm = oi; }
    }
    if (jm >= n) { jm = j; maxval = 0.0; }

    // ---- B. the pivot column and the column it displaces go to LDS (8 owner lanes each)
    const int wm = (jm & 63) >> 3;  // owner wave of the pivot position
    // Wave-uniform control flow only: every lane of the owner wave stores, the 56 lanes of the
    // other column groups into a dummy area (lane-divergent bulk copies of the register tile make
    // the register allocator spill it).  Two passes: one wave may own both columns.
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass == 0 ? jm : j;
        if (!(pass == 1 && jm == j) && w == ((p & 63) >> 3)) {
            double *dst = (cg == (p & 7)) ? (pass == 0 ? colP : colJ) : dummy;
            const int sp = p >> 6;
            if (sp == 3) {
#pragma unroll
                for (int k = 0; k < 32; ++k) dst[rg + 8 * k] = x3[k];
            } else if (sp == 2) {
#pragma unroll
                for (int k = 0; k < 32; ++k) dst[rg + 8 * k] = x2[k];
            } else if (sp == 1) {
                for (int k = 0; k < 32; ++k) dst[rg + 8 * k] = l1[rg + 8 * k];
            } else {
#pragma unroll 16
                for (int k = 0; k < 32; ++k) {
                    const int r = rg + 8 * k;
                    dst[r] = r < n ? g0[r] : 0.0;
                }
            }
        }
    }
    __syncthreads();
    // reflector (UDT.jl:133-148); column j is final: R above the diagonal, v below
    const double xi1 = colP[j];
    double tj = 0.0, nu = 0.0, xi = 1.0;
    if (maxval != 0.0) {
        nu = copysign(sqrt(maxval), xi1);
        xi = xi1 + nu;
        tj = xi / nu;
    }
    if (tid < 256) {
        const int r = tid;
        double vr = 0.0;
        if (r < n) {
            const double cv = colP[r];
            double outv = cv;
            vr = (r == j) ? 1.0 : 0.0;
            if (maxval != 0.0) {
                if (r == j) outv = -nu;
                else if (r > j) { outv = cv / xi; vr = outv; }
            }
            // position j is dead from now on: its storage (wherever it was) is never read again,
            // so the finished column goes straight to the output matrix
            A[r + (long)n * j] = outv;
        }
        vbuf[(r & 7) * QT_VS + (r >> 3)] = vr;  // permuted: slice of row group r & 7
    }
    if (tid == 0) {
        tau[j] = tj;
        const int t = pv[j];
        pv[j] = pv[jm];
        pv[jm] = t;
    }
    if (jm != j && w == wm) {  // the displaced column moves into the storage of position jm
        const bool mine = cg == (jm & 7);
        const int sp = jm >> 6;
        if (sp == 3) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { const double cc = colJ[rg + 8 * k]; x3[k] = mine ? cc : x3[k]; }
        } else if (sp == 2) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { const double cc = colJ[rg + 8 * k]; x2[k] = mine ? cc : x2[k]; }
        } else if (mine) {
            if (sp == 1) {
                for (int k = 0; k < 32; ++k) l1[rg + 8 * k] = colJ[rg + 8 * k];
            } else {
                for (int k = 0; k < 32; ++k) {
                    const int r = rg + 8 * k;
                    if (r < n) g0[r] = colJ[r];
                }
            }
        }
    }
    __syncthreads();

    // ---- C. apply H_j to the trailing columns (reflectorApply!, UDT.jl:32-50); norms for step j+1.
    // Finished rows inside a live row block have v = 0 (no effect on dot/update) and are masked in the norm.
    const double *vq = vbuf + rg * QT_VS;
    if (8 * w + 7 + 192 > j) qt_reg_col<KB0>(x3, vq, nrmp, rg, 192 + pb, j, tj);
    if (KB0 < 6 && 8 * w + 7 + 128 > j) qt_reg_col<KB0>(x2, vq, nrmp, rg, 128 + pb, j, tj);
    if (KB0 < 4 && 8 * w + 7 + 64 > j) {  // slot 1: LDS resident, rolled loops
        double d = 0.0;
#pragma unroll 8
        for (int k = 4 * KB0; k < 32; ++k) d += vq[k] * l1[rg + 8 * k];
        const double wv = (64 + pb > j) ? sum8(d) * tj : 0.0;
        double nr = 0.0;
#pragma unroll 8
        for (int k = 4 * KB0; k < 32; ++k) {
            const int r = rg + 8 * k;
            const double y = l1[r] - vq[k] * wv;
            l1[r] = y;
            nr += (r > j ? 1.0 : 0.0) * (y * y);
        }
        nrmp[rg * 256 + 64 + pb] = nr;
    }
    if (KB0 < 2 && 8 * w + 7 > j) {       // slot 0: in place in global memory (L2), the column is read twice
        double d = 0.0;
        if (FULLN) {
            // QT_GB requests in flight at a time, by hand: at its register limit the compiler requests, waits and
            // accumulates one element after the other (32 trips to the L2 per pass)
#pragma unroll 1
            for (int k0 = 4 * KB0; k0 < 32; k0 += QT_GB) {
                double t[QT_GB];
#pragma unroll
                for (int i = 0; i < QT_GB; ++i) t[i] = __builtin_nontemporal_load(g0 + rg + 8 * (k0 + i));
#pragma unroll
                for (int i = 0; i < QT_GB; ++i) d = __builtin_fma(vq[k0 + i], t[i], d);
            }
        } else {
#pragma unroll 16
            for (int k = 4 * KB0; k < 32; ++k) {
                const int r = rg + 8 * k;
                d += vq[k] * (r < n ? g0[r] : 0.0);
            }
        }
        const double wv = (pb > j) ? sum8(d) * tj : 0.0;
        double nr = 0.0;
        if (FULLN) {
#pragma unroll 1
            for (int k0 = 4 * KB0; k0 < 32; k0 += QT_GB) {
                double t[QT_GB];
#pragma unroll
                for (int i = 0; i < QT_GB; ++i) t[i] = g0[rg + 8 * (k0 + i)];
#pragma unroll
                for (int i = 0; i < QT_GB; ++i) {
                    const int r = rg + 8 * (k0 + i);
                    const double y = t[i] - vq[k0 + i] * wv;
                    g0[r] = y;
                    nr += (r > j ? 1.0 : 0.0) * (y * y);
                }
            }
        } else {
#pragma unroll 16
            for (int k = 4 * KB0; k < 32; ++k) {
                const int r = rg + 8 * k;
                if (r < n) {
                    const double y = g0[r] - vq[k] * wv;
                    g0[r] = y;
                    nr += (r > j ? 1.0 : 0.0) * (y * y);
                }
            }
        }
        nrmp[rg * 256 + pb] = nr;
    }
    __syncthreads();
}

template <bool FULLN>
__global__ __launch_bounds__(QT_THREADS) void qr_tile256_kernel(int n_rt, int nsteps, double *__restrict__ Aall, long strideA,
                                                               double *__restrict__ tauall,
                                                               int *__restrict__ pivall,
                                                               const double *__restrict__ srcall, long strideSrc,
                                                               int *guard, int guard_val, double *__restrict__ Xall,
                                                               long strideX)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (guard && guard[0] != guard_val) return;  // fallback use, see qr_pivot_kernel
    const int n = FULLN ? 256 : n_rt;
    if (srcall) {
        const double *__restrict__ src = srcall + (long)blockIdx.x * strideSrc;
        double *__restrict__ dst = Aall + (long)blockIdx.x * strideA;
        for (long i = threadIdx.x; i < (long)n * n; i += QT_THREADS) dst[i] = src[i];
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&guard[1], 1);
        __syncthreads();
    }
    __shared__ double cand_v[8];
    __shared__ int cand_i[8];
    double *Lc = sm;                        // [64][QT_LSTRIDE] positions 64..127
    double *colP = Lc + 64 * QT_LSTRIDE;    // [256] pivot column
    double *colJ = colP + 256;              // [256] column that sat at position j
    double *vbuf = colJ + 256;              // [8][QT_VS] reflector, permuted by row group
    double *nrmp = vbuf + 8 * QT_VS;        // [8][256] norm partials per row group
    double *dummy = nrmp + 8 * 256;         // [256] sink for the non-owner lanes of a column copy
    int *pv = (int *)(dummy + 256);         // [256] pivot vector
    const int unit = blockIdx.x;
    double *__restrict__ A = Aall + (long)unit * strideA;
    double *__restrict__ tau = tauall + (long)unit * n;
    int *__restrict__ piv = pivall + (long)unit * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = lane & 7, cg = lane >> 3;
    const int pb = 8 * w + cg;

    double x2[32], x3[32];
    {   // load the on-chip part of the matrix and the initial norm partials
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int r = rg + 8 * k;
            const bool rok = r < n;
            const double a0 = (rok && pb < n) ? A[r + (long)n * pb] : 0.0;
            const double a1 = (rok && pb + 64 < n) ? A[r + (long)n * (pb + 64)] : 0.0;
            const double a2 = (rok && pb + 128 < n) ? A[r + (long)n * (pb + 128)] : 0.0;
            const double a3 = (rok && pb + 192 < n) ? A[r + (long)n * (pb + 192)] : 0.0;
            Lc[pb * QT_LSTRIDE + r] = a1;
            x2[k] = a2;
            x3[k] = a3;
            n0 += a0 * a0; n1 += a1 * a1; n2 += a2 * a2; n3 += a3 * a3;
        }
        nrmp[rg * 256 + pb] = n0;
        nrmp[rg * 256 + pb + 64] = n1;
        nrmp[rg * 256 + pb + 128] = n2;
        nrmp[rg * 256 + pb + 192] = n3;
    }
    if (tid < 256) pv[tid] = tid;
    __syncthreads();

    // eight regions of 32 steps, each with its compile-time first live row block
#define QT_REGION(REG)                                             \
    for (int j = 32 * (REG); j < nsteps && j < 32 * (REG) + 32; ++j) \
        qt_step<(REG), FULLN>(n, j, A, tau, Lc, colP, colJ, vbuf, nrmp, dummy, pv, cand_v, cand_i, x2, x3);
    QT_REGION(0) QT_REGION(1) QT_REGION(2) QT_REGION(3) QT_REGION(4) QT_REGION(5) QT_REGION(6) QT_REGION(7)
#undef QT_REGION
    if (tid < n) piv[tid] = pv[tid];
    // two-phase form (nsteps = 128 of n = 256, used when there are too many matrices for the cooperative kernel): the
    // trailing columns are exactly the register-resident positions 128..255; they go to X by ORIGINAL column id, all
    // rows, and qr_tail_kernel<128, 4> takes over with the position table just written
    if (Xall && nsteps == 128 && n == 256) {
        double *__restrict__ X = Xall + (long)unit * strideX;
        double *__restrict__ c2 = X + 256l * pv[128 + pb] + rg, *__restrict__ c3 = X + 256l * pv[192 + pb] + rg;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            c2[8 * k] = x2[k];
            c3[8 * k] = x3[k];
        }
    }
}

// ---------------------------------------------------------------------------
// Cooperative variant for n <= 256: EIGHT workgroups factor one matrix, so that the 32 matrices of
// the benchmark batch occupy all 256 CUs instead of 32, and every workgroup's share (32 columns)
// lives in registers: 256 threads, thread (wave w, lane l): row group rg = l & 7, column group
// cg = l >> 3 holds rows rg + 8k (k = 0..31) of ORIGINAL column c = part + 8 (8 w + cg).
// Pivoting is logical (no column ever moves between workgroups): every workgroup keeps the same
// replicated position tables pos[column], colat[position], so the reference's tie-break (first
// maximum in position order, UDT.jl:151-168) and its swap bookkeeping (UDT.jl:219-231) are kept.
//
// One hand-off per step through a global mailbox, with SELF-VALIDATING packets (the LL protocol
// of collective libraries): every double travels as two 8-byte words {low half, tag} and
// {high half, tag}, each written with one agent-scope relaxed (sc1, write-through) 8-byte atomic
// store and read with one sc1 8-byte atomic load, so a reader that sees the tag of this step has
// the payload of this step -- no store drain, no separate flag, no second round trip for the
// header.  Each workgroup publishes the raw column of its best live column plus a header
// {norm, position, column id}; 8 lanes poll the 8 headers, then every thread polls its element of
// the winning column.  Tags are epoch*1024 + step + 1 (no reset between launches; memory starts
// zeroed and epoch >= 1), mailboxes are double buffered by step parity (the skew between
// workgroups is at most one step, because a step cannot finish without everybody's header).
// Every spin is bounded; a timeout raises a flag that the host reports.
// workgroup barrier of the step loop: LDS traffic only is ordered (s_waitcnt lgkmcnt(0)); the packet and output stores
// of the step stay in flight (__syncthreads() would also wait for vmcnt = 0, i.e. for every store's acknowledgement).
// Packets validate themselves, so nothing here relies on the order or completion of global stores.
#define QC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
constexpr int QC_PARTS = 8;
constexpr int QC_MB = 264;          // packets (16 B) per mailbox slot: 256 column entries + header
constexpr unsigned QC_SPIN_LIMIT = 400000u;  // about 0.2 s of polling: far beyond any legitimate co-residency delay

typedef unsigned long long qc_word;
__device__ __forceinline__ void qc_put(qc_word *slot, double v, unsigned tag_lo, unsigned tag_hi)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(slot, (bits & 0xffffffffull) | ((unsigned long long)tag_lo << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(slot + 1, (bits >> 32) | ((unsigned long long)tag_hi << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// same payload with workgroup-scope stores: they stay in the XCD's L2 (MI355X_MICROARCH.md, visibility
// table: plain / sc0 stores KEEP the line, sc1 stores DROP it and every reader pays the cross-XCD
// rate).  Only other CUs of the SAME XCD can see them, so this form is used only after the eight
// workgroups of a matrix have verified at run time that they share one XCD (see the kernel).
__device__ __forceinline__ void qc_put_local(qc_word *slot, double v, unsigned tag_lo, unsigned tag_hi)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(slot, (bits & 0xffffffffull) | ((unsigned long long)tag_lo << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(slot + 1, (bits >> 32) | ((unsigned long long)tag_hi << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spins until both halves carry their tags (tag_hi is compared under mask_hi); false on timeout
__device__ __forceinline__ bool qc_get(const qc_word *slot, unsigned tag_lo, unsigned tag_hi, unsigned mask_hi,
                                       double &v, unsigned &hi_word)
{
    for (unsigned spins = 0; spins < QC_SPIN_LIMIT; ++spins) {
        const unsigned long long a = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(a >> 32) == tag_lo && (((unsigned)(b >> 32)) & mask_hi) == tag_hi) {
            v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
            hi_word = (unsigned)(b >> 32);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// wave-wide reductions on DPP (no LDS, no ds_bpermute): results are wave-uniform
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_self_f64(double x)
{
    // lanes without a valid source (or in masked rows) get their own value back
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, BANK_MASK, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_self_u32(unsigned x)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, BANK_MASK, false);
}
// maximum over the wave, returned wave-uniform.  FROM = 8: the value is already equal inside
// groups of 8 consecutive lanes (only lane bits 3..5 need combining)
template <int FROM>
__device__ __forceinline__ double wave_max_f64(double v)
{
    if (FROM <= 1) v = fmax(v, dpp_self_f64<0x111>(v));  // row_shr:1
    if (FROM <= 2) v = fmax(v, dpp_self_f64<0x112>(v));  // row_shr:2
    if (FROM <= 4) v = fmax(v, dpp_self_f64<0x114>(v));  // row_shr:4
    v = fmax(v, dpp_self_f64<0x118>(v));                 // row_shr:8  -> lane 15 of each row
    v = fmax(v, dpp_self_f64<0x142, 0xa>(v));            // row_bcast:15 into rows 1, 3
    v = fmax(v, dpp_self_f64<0x143, 0xc>(v));            // row_bcast:31 into rows 2, 3 -> lane 63
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int FROM>
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    if (FROM <= 1) v = min(v, dpp_self_u32<0x111>(v));
    if (FROM <= 2) v = min(v, dpp_self_u32<0x112>(v));
    if (FROM <= 4) v = min(v, dpp_self_u32<0x114>(v));
    v = min(v, dpp_self_u32<0x118>(v));
    v = min(v, dpp_self_u32<0x142, 0xa>(v));
    v = min(v, dpp_self_u32<0x143, 0xc>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// sqrt(x) and 1/sqrt(x) together (coupled Goldschmidt iterations from v_rsq_f64, then one correction of the root);
// x > 0 and far from the ends of the exponent range
__device__ __forceinline__ void qb_sqrt_rsqrt(double x, double &root, double &rroot)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    root = __builtin_fma(d, h, g);
    rroot = h + h;
}
// 1/x from v_rcp_f64 and two Newton steps (no scaling, no special cases: x is a column norm scale)
__device__ __forceinline__ double qb_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// reflectorApply! (UDT.jl:32-50) on one register-resident column, followed by its fresh squared norm.
// KB0 = first live block of 8 rows: in the region of steps 32 R .. 32 R + 31 rows below 32 R are finished
// (the reflector is zero there), so blocks k < 4 R cost nothing at compile time.
template <int KB0>
__device__ __forceinline__ double qc_apply(double (&x)[32], const double *vq, double rcp, double trr, int j, int rg)
{
    double2 vv[16];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = KB0; k < 32; k += 2) {
        vv[k >> 1] = *reinterpret_cast<const double2 *>(vq + k);
        d0 += vv[k >> 1].x * x[k];
        d1 += vv[k >> 1].y * x[k + 1];
    }
    // vq holds the UNSCALED reflector u = xi v (u_j = xi): H x = x - u (tj / xi^2) (u' x)
    const double wv = (sum8(d0 + d1) * rcp) * trr;
    double n0 = 0.0, n1 = 0.0;
#pragma unroll
    for (int k = KB0; k < 32; k += 2) {
        const double y0 = x[k] - vv[k >> 1].x * wv, y1 = x[k + 1] - vv[k >> 1].y * wv;
        x[k] = y0;
        x[k + 1] = y1;
#ifndef QR_NONORM_EXPERIMENT
        if (k < KB0 + 4) {  // only the region's own 32 rows can be <= j
            n0 += (rg + 8 * k > j ? 1.0 : 0.0) * (y0 * y0);
            n1 += (rg + 8 * (k + 1) > j ? 1.0 : 0.0) * (y1 * y1);
        } else {
            n0 += y0 * y0;
            n1 += y1 * y1;
        }
#endif
    }
#ifdef QR_NONORM_EXPERIMENT  // timing experiment only (wrong pivots): what a step costs without the fresh norms
    return wv * 1e-300 + 1.0;
#else
    return sum8(n0 + n1);
#endif
}

__global__ __launch_bounds__(256) void qr_coop_kernel(int n, int n_units, double *__restrict__ Aall, long strideA,
                                                     double *__restrict__ tauall, int *__restrict__ pivall,
                                                     double *mailbox,
                                                     unsigned long long epoch, int *fb, int force_sc1,
                                                     double *__restrict__ Wall, long strideW, int force_timeout,
                                                     int nsteps, double *__restrict__ Xall, long strideX)
{
    // my best column, permuted by row group like vperm: row r at (r & 7) * QT_VS + (r >> 3), so that an owner
    // lane (rows rg + 8k) stores pairs of consecutive k with 16-byte writes
    __shared__ __attribute__((aligned(16))) double colbuf[8 * QT_VS];
    __shared__ __attribute__((aligned(16))) double vperm[8 * QT_VS];
    __shared__ __attribute__((aligned(16))) double wc[4][2];   // per wave: {norm, bits(pos | col << 32)}
    __shared__ __attribute__((aligned(16))) double win[4];     // the step's pivot: {norm, bits(pos | col << 32), part, -}
    __shared__ int colat[256];  // position -> column (the inverse, pos[my column], lives in a register)
    __shared__ int s_abort;
    __shared__ int xcc_seen[QC_PARTS];

    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int unit = (seq / QC_PARTS) * 8 + xcd, part = seq % QC_PARTS;
    if (unit >= n_units) return;  // all parts of a missing unit leave together
    // The factored matrix goes to W, the input A is only read: a hand-off time-out therefore leaves the input intact
    // and the guarded single-workgroup kernel launched behind this one redoes the factorisation (fb[0] = epoch).
    if (force_timeout == 1) {  // test hook (DQMC_QR_FORCE_TIMEOUT=1): behave like a launch whose hand-offs timed out
        if (threadIdx.x == 0) atomicExch(&fb[0], (int)(epoch & 0x7fffffffull));
        return;
    }
    const double *__restrict__ A = Aall + (long)unit * strideA;
    double *__restrict__ Wo = Wall + (long)unit * strideW;
    double *__restrict__ tau = tauall + (long)unit * n;
    int *__restrict__ piv = pivall + (long)unit * n;
    qc_word *mb_unit = reinterpret_cast<qc_word *>(mailbox) + (long)unit * 2 * QC_PARTS * QC_MB * 2;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = lane & 7, cg = lane >> 3;
    const int c = part + 8 * (8 * w + cg);  // my original column
    double x[32];
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int r = rg + 8 * k;
        const double a = (r < n && c < n) ? A[r + (long)n * c] : 0.0;
        x[k] = a;
        nrm += a * a;
    }
    nrm = sum8(nrm);  // full squared norm of column c, in all 8 lanes of the column
    int mypos = c;  // pos[c]
    colat[tid] = tid;
    if (tid == 0) s_abort = 0;
    __syncthreads();

    // ---- placement check (once per launch): HIP promises nothing about workgroup -> XCD placement, so
    // the eight parts exchange their XCC ids through agent-scope (sc1) packets and use the L2-resident
    // store form only if all eight sit on one XCD; otherwise every packet is written through (sc1)
    {
        const unsigned tag0 = (unsigned)(epoch * 1024ull);   // step tags start at epoch*1024 + 1
        const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        qc_word *pre = mb_unit;  // packet 260 of the parity-0 slots is used by nothing else
        if (tid == 0) qc_put(pre + ((long)part * QC_MB + 260) * 2, (double)my_xcc, tag0, tag0);
        if (tid < QC_PARTS) {
            double v = -1.0;
            unsigned hw;
            if (!qc_get(pre + ((long)tid * QC_MB + 260) * 2, tag0, tag0, 0xffffffffu, v, hw)) s_abort = 1;
            xcc_seen[tid] = (int)v;
        }
        __syncthreads();
    }
    bool same_xcd = true;
#pragma unroll
    for (int q = 1; q < QC_PARTS; ++q) same_xcd = same_xcd && xcc_seen[q] == xcc_seen[0];
    if (force_sc1) same_xcd = false;

    for (int j = 0; j < nsteps; ++j) {
        // test hook (DQMC_QR_FORCE_TIMEOUT=step:<j>): part 3 drops out of the hand-off at step j, as a workgroup that
        // lost its CU would; the other parts run into the bounded spins, raise fb[0] and leave the input to the
        // guarded kernel behind this launch
        if (force_timeout >= 2 && part == 3 && j == force_timeout - 2) return;
        const int par = j & 1;
        const unsigned tag = (unsigned)(epoch * 1024ull + (unsigned long long)j + 1ull);
        // the column now at position j: read here, three barriers before thread 0 rewrites the table
        const int cj = colat[j];
        // ---- my workgroup's best live column: larger norm first, then smaller position
        {
            const bool live = c < n && mypos >= j;
            const double bn = wave_max_f64<8>(live ? nrm : -1.0);
            const unsigned mp = live ? (unsigned)mypos : 0xffffffffu;
            const unsigned bp = wave_min_u32<8>((live && nrm == bn) ? mp : 0xffffffffu);
            int bc = -1;
            if (bn >= 0.0) bc = __builtin_amdgcn_readlane(c, __ffsll((long long)__ballot(live && mp == bp)) - 1);
            if (lane == 0) {
                wc[w][0] = bn;
                wc[w][1] = __longlong_as_double((long long)(bp & 0x7fffffffu) | ((long long)(unsigned)bc << 32));
            }
        }
        QC_BARRIER();
        double lbn = -1.0;
        int lbp = 0x7fffffff, lbc = -1;
        {
            double2 cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cq[q] = *reinterpret_cast<const double2 *>(wc[q]);  // four loads in flight
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long pc = __double_as_longlong(cq[q].y);
                const int qp = (int)(pc & 0x7fffffff), qc = (int)(pc >> 32);
                // bitwise logic: with || and && the compiler builds a chain of scalar branches around each LDS read
                const bool better = (cq[q].x > lbn) | ((cq[q].x == lbn) & (qp < lbp));
                lbn = better ? cq[q].x : lbn;
                lbp = better ? qp : lbp;
                lbc = better ? qc : lbc;
            }
        }
        // ---- its owner lanes put the column into LDS (measured: letting every wave extract its own candidate before
        // the first barrier instead, as qr_tail_kernel does, saves this barrier but costs 140 ns per step here)
        if (lbc >= 0 && w == ((lbc >> 3) >> 3)) {
            if (cg == ((lbc >> 3) & 7)) {  // exec-masked: the other lane groups issue no LDS writes at all
                double2 *dst = reinterpret_cast<double2 *>(colbuf + rg * QT_VS);
#pragma unroll
                for (int k = 0; k < 32; k += 2) dst[k >> 1] = make_double2(x[k], x[k + 1]);
            }
        }
        QC_BARRIER();
        // ---- publish: raw column + header as tagged packets; nothing to wait for
        {
            qc_word *mb = mb_unit + (long)(par * QC_PARTS + part) * QC_MB * 2;
            const double pv = lbc >= 0 ? colbuf[(tid & 7) * QT_VS + (tid >> 3)] : 0.0;
            // header: packet 256 carries the norm, packet 258 {position, column id}; every word has the full tag
            const double meta = __longlong_as_double((long long)(unsigned)lbp | ((long long)(unsigned)lbc << 32));
            // packet 262: element j of the candidate column, so that the reflector scalars do not have to wait
            // for the column itself
            const double pivot_elem = (tid == 0 && lbc >= 0) ? colbuf[(j & 7) * QT_VS + (j >> 3)] : 0.0;
            if (same_xcd) {
                qc_put_local(mb + 2 * tid, pv, tag, tag);
                if (tid == 0) {
                    qc_put_local(mb + 2 * 256, lbn, tag, tag);
                    qc_put_local(mb + 2 * 258, meta, tag, tag);
                    qc_put_local(mb + 2 * 262, pivot_elem, tag, tag);
                }
            } else {
                qc_put(mb + 2 * tid, pv, tag, tag);
                if (tid == 0) {
                    qc_put(mb + 2 * 256, lbn, tag, tag);
                    qc_put(mb + 2 * 258, meta, tag, tag);
                    qc_put(mb + 2 * 262, pivot_elem, tag, tag);
                }
            }
        }
        // ---- collect: 16 lanes of wave 0 poll the 8 headers (lane l: packet 256 + 2 (l & 1) of part l >> 1) and
        // pick the step's pivot: largest norm, then smallest position (UDT.jl:151-168); only the result
        // goes through LDS
        if (w == 0) {
            double hv = -1.0;
            if (lane < 3 * QC_PARTS) {  // lanes 0..15: norm / {pos, col} of part lane >> 1; lanes 16..23: its element j
                unsigned hw;
                const int hp = lane < 2 * QC_PARTS ? (lane >> 1) : lane - 2 * QC_PARTS;
                const int hk = lane < 2 * QC_PARTS ? 256 + 2 * (lane & 1) : 262;
                const qc_word *mb = mb_unit + (long)(par * QC_PARTS + hp) * QC_MB * 2;
                if (!qc_get(mb + 2 * hk, tag, tag, 0xffffffffu, hv, hw)) s_abort = 1;
            }
            // lane 2q holds the norm of part q, lane 2q+1 its {position, column}: give both to both lanes of the
            // pair, then reduce across the wave on DPP (largest norm, then smallest position; lanes >= 16 idle)
            const double other = dpp_self_f64<0xB1>(hv);  // quad_perm [1,0,3,2]
            const double qn = (lane < 2 * QC_PARTS) ? ((lane & 1) ? other : hv) : -1.0;
            const double qmeta = (lane & 1) ? hv : other;
            const unsigned qp = (unsigned)__double2loint(qmeta);
            const int qc = __double2hiint(qmeta);
            const double bestn = wave_max_f64<2>(qn);
            const bool cand = qn >= 0.0 && qn == bestn;
            const unsigned bestp_u = wave_min_u32<2>(cand ? qp : 0xffffffffu);
            int bestp = 0x7fffffff, bestc = -1, bestq = 0;
            if (bestn >= 0.0) {
                const int wl = __ffsll((long long)__ballot(cand && qp == bestp_u)) - 1;
                bestp = (int)bestp_u;
                bestc = __builtin_amdgcn_readlane(qc, wl);
                bestq = wl >> 1;
            }
            const double best_xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(hv), 2 * QC_PARTS + bestq),
                                                    __builtin_amdgcn_readlane(__double2loint(hv), 2 * QC_PARTS + bestq));
            if (lane == 0) {
                win[0] = bestn;
                win[1] = __longlong_as_double((long long)bestp | ((long long)(unsigned)bestc << 32));
                win[2] = (double)bestq;
                win[3] = best_xj;
            }
        }
        QC_BARRIER();
        double maxval, xi1;
        int jm, cm, wpart;
        {
            const double2 w01 = *reinterpret_cast<const double2 *>(win);
            const long long pc = __double_as_longlong(w01.y);
            const double2 w23 = *reinterpret_cast<const double2 *>(win + 2);
            maxval = w01.x; jm = (int)(pc & 0x7fffffff); cm = (int)(pc >> 32); wpart = (int)w23.x;
            xi1 = w23.y;
        }
        if (cm < 0) { cm = cj; jm = j; maxval = 0.0; wpart = cm & 7; }  // nothing live: cannot happen for j < n
        // ---- reflector scalars (UDT.jl:133-148) from the header alone; they overlap the column fetch below
        // (branch-free: a zero column, maxval == 0, keeps tau = 0 and the column as it is)
        const bool nz = maxval != 0.0;
        double rootn, rrootn;
        qb_sqrt_rsqrt(nz ? maxval : 1.0, rootn, rrootn);
        const double nu = nz ? copysign(rootn, xi1) : 1.0;
        const double xi = nz ? xi1 + nu : 1.0;
        const double tj = nz ? __builtin_fma(fabs(xi1), rrootn, 1.0) : 0.0;  // xi / nu
        const double rcp = qb_rcp(xi), trr = tj * rcp;
        // ---- the winning column (tagged packets), output column j
        double cv = 0.0;
        {
            unsigned hw;
            if (!qc_get(mb_unit + (long)(par * QC_PARTS + wpart) * QC_MB * 2 + 2 * tid, tag, tag, 0xffffffffu, cv, hw))
                s_abort = 1;
        }
        {
            const int r = tid;
            double outv = (nz && r == j) ? -nu : ((nz && r > j) ? cv * rcp : cv);
            double vr = (r == j) ? xi : ((nz && r > j) ? cv : 0.0);  // unscaled: u = xi v
            if (r >= n) vr = 0.0;
            vperm[(r & 7) * QT_VS + (r >> 3)] = vr;
            if (part == wpart && r < n) Wo[r + (long)n * j] = outv;  // the owner writes the finished column
        }
        // swap positions j <-> jm (UDT.jl:219-231): every thread tracks its own column, thread 0 the table
        {
            if (c == cm) mypos = j;
            else if (c == cj) mypos = jm;  // (cj == cm when jm == j: covered by the first branch)
            if (tid == 0) {
                if (part == wpart) tau[j] = tj;
                colat[j] = cm;
                if (jm != j) colat[jm] = cj;
            }
        }
        QC_BARRIER();
        if (s_abort) break;
        // ---- apply H_j to my column if it is still live (reflectorApply!, UDT.jl:32-50); fresh norm
        if (c < n && mypos > j) {
            const double *vq = vperm + rg * QT_VS;
            switch (j >> 5) {  // wave-uniform: region of 32 steps -> first live row block
            case 0: nrm = qc_apply<0>(x, vq, rcp, trr, j, rg); break;
            case 1: nrm = qc_apply<4>(x, vq, rcp, trr, j, rg); break;
            case 2: nrm = qc_apply<8>(x, vq, rcp, trr, j, rg); break;
            case 3: nrm = qc_apply<12>(x, vq, rcp, trr, j, rg); break;
            case 4: nrm = qc_apply<16>(x, vq, rcp, trr, j, rg); break;
            case 5: nrm = qc_apply<20>(x, vq, rcp, trr, j, rg); break;
            case 6: nrm = qc_apply<24>(x, vq, rcp, trr, j, rg); break;
            default: nrm = qc_apply<28>(x, vq, rcp, trr, j, rg); break;
            }
        }
        // (pos/colat are rewritten by thread 0 only after the barriers of the next step)
    }
    if (s_abort) {
        if (tid == 0) atomicExch(&fb[0], (int)(epoch & 0x7fffffffull));
        return;
    }
    __syncthreads();
    if (part == 0 && tid < n) piv[tid] = colat[tid];
    // two-phase form (nsteps < n): the live columns go to X by ORIGINAL column id, all rows (the finished R entries
    // above row nsteps included); qr_tail_kernel continues from there with the position table just written
    if (nsteps < n && c < n && mypos >= nsteps) {
        double *__restrict__ Xc = Xall + (long)unit * strideX + (long)n * c;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (rg + 8 * k < n) Xc[rg + 8 * k] = x[k];
    }
}

#ifdef QB_STAMPS
__device__ long long *qb_stamp_ptr = nullptr;  // [wave 8][step 192][point 8] of unit 0 (diagnostic build only)
#endif

// ---------------------------------------------------------------------------
// Tail of the factorisation for n == 256.  After QB_J0 cooperative steps the trailing (256 - QB_J0)^2 block fits
// the registers of ONE CU, and a step there needs no hand-off between workgroups at all: two LDS-only barriers
// per step instead of two mailbox round trips through L2.  One workgroup of 512 threads per matrix; slot
// s = (position at hand-over) - QB_J0 lives in thread group g = s % 64 (wave g / 8, lanes 8 (g % 8) .. + 7) as
// column s / 64 of that group, lane rg holding rows QB_J0 + rg + 8 k.  Pivoting stays logical (slots never move),
// with the reference's rule: largest norm of the updated trailing column, first in position order
// (UDT.jl:151-168), swap bookkeeping as UDT.jl:219-231, reflector as UDT.jl:133-148 / :32-50.
// Rows < QB_J0 of the trailing columns are final when the hand-over happens; they are copied X -> W at the end,
// when the positions are known.
// Geometry of a tail kernel that takes over after J0 steps with NW waves: M = 256 - J0 trailing rows and columns,
// 8 NW thread groups of 8 lanes, CPT = M / (8 NW) columns per group, KR = M / 8 rows per lane.
//   <128, 4>: 128 x 128 on ONE wave per SIMD (the step is bound by VALU issue: two waves per SIMD halve each other's rate
//             in the replicated selection / scalar work) - the default;  <64, 8>: 192 x 192 on two waves per SIMD
template <int J0, int NW>
struct QbGeo {
    static constexpr int M = 256 - J0, KR = M / 8, CPT = M / (8 * NW), VS = KR + 2, THREADS = 64 * NW, REGIONS = M / 32;
    static_assert(M % 32 == 0 && M % (8 * NW) == 0 && KR % 4 == 0, "qr_tail_kernel geometry");
    static_assert(CPT == 3 || CPT == 4, "qb_step: extraction copies are written out for 3 or 4 columns per group");
};

#ifdef QB_STAMPS
#define QB_STAMP(P)                                                                                      \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && qb_stamp_ptr)                                      \
        qb_stamp_ptr[((threadIdx.x >> 6) * 192 + t) * 8 + (P)] = __builtin_amdgcn_s_memtime();
#else
#define QB_STAMP(P)
#endif
template <int J0, int NW>
struct QbShared {
    using G = QbGeo<J0, NW>;
    // candidate columns of the waves, double-buffered by step parity (a wave is at most one barrier ahead):
    // row r of wave q's candidate at colbuf[par][q][((r - J0) & 7) * VS + ((r - J0) >> 3)]
    __attribute__((aligned(16))) double colbuf[2][NW][8 * G::VS];
    __attribute__((aligned(16))) double wc[2][NW][2];  // per wave: {norm, bits(position | slot << 32)}
    int posmap[G::M];                                  // (final) position - J0 -> original column
    int colid[G::M];                                   // slot -> original column
};

// candidate b replaces a: larger norm, then smaller position (bitwise logic: no branches)
__device__ __forceinline__ void qb_merge(double &an, int &ap, int &as, double bn, int bp, int bs)
{
    const bool better = (bn > an) | ((bn == an) & (bp < ap));
    an = better ? bn : an;
    ap = better ? bp : ap;
    as = better ? bs : as;
}

// One step, ONE workgroup barrier: every wave extracts its own best live column before the barrier, so that the
// winner's column is already in LDS when the candidates meet.
// KB0 = first live block of 8 rows (4 per region of 32 steps)
template <int J0, int NW, int KB0>
__device__ __forceinline__ void qb_step(int t, double (&x)[QbGeo<J0, NW>::CPT][QbGeo<J0, NW>::KR],
                                        double (&nrm)[QbGeo<J0, NW>::CPT], int (&mypos)[QbGeo<J0, NW>::CPT],
                                        QbShared<J0, NW> &sm, double *__restrict__ Wo, double *__restrict__ tau)
{
    using G = QbGeo<J0, NW>;
    constexpr int CPT = G::CPT, KR = G::KR, VS = G::VS, NG = 8 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = lane & 7, cg = lane >> 3, g = 8 * w + cg;
    const int j = J0 + t, par = t & 1;
    QB_STAMP(0)
    // ---- best live column of my wave, extracted by its owner lanes
    {
        double bn = -1.0;
        int bp = 0x7fffffff, bs = -1;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const bool better = (mypos[c] >= j) & ((nrm[c] > bn) | ((nrm[c] == bn) & (mypos[c] < bp)));
            bn = better ? nrm[c] : bn;
            bp = better ? mypos[c] : bp;
            bs = better ? g + NG * c : bs;
        }
        const double wn = wave_max_f64<8>(bn);
        const bool eq = (bn == wn) & (bn >= 0.0);
        unsigned long long m = __ballot(eq);
        int first = m ? __ffsll((long long)m) - 1 : 0;
        if ((m >> (first & ~7)) >> 8) {  // equal norms in several columns (wave-uniform, rare): smallest position
            const unsigned wp = wave_min_u32<8>(eq ? (unsigned)bp : 0xffffffffu);
            m = __ballot(eq & ((unsigned)bp == wp));
            first = __ffsll((long long)m) - 1;
        }
        const int ws = m ? __builtin_amdgcn_readlane(bs, first) : -1;
        const int wp = __builtin_amdgcn_readlane(bp, first);
        QB_STAMP(1)
        if (ws >= 0 && cg == (ws & 7)) {
            double2 *dst = reinterpret_cast<double2 *>(sm.colbuf[par][w] + rg * VS);
            const int oc = ws / NG;
            // separate copies with static register indices; the distinct asm comments keep the compiler from
            // merging them into one copy with a run-time column index (which would put x[][] into scratch)
#define QB_EXTRACT(C)                                                                               \
    {                                                                                               \
        asm volatile("; extract column " #C);                                                       \
        _Pragma("unroll") for (int k = 0; k < KR; k += 2) dst[k >> 1] = make_double2(x[C][k], x[C][k + 1]); \
        asm volatile("; extracted column " #C);                                                     \
    }
            if (oc == 0) QB_EXTRACT(0)
            else if (oc == 1) QB_EXTRACT(1)
            else if (CPT == 3 || oc == 2) QB_EXTRACT(2)
            else QB_EXTRACT(CPT - 1)
#undef QB_EXTRACT
        }
        if (lane == 0) {
            *reinterpret_cast<double2 *>(sm.wc[par][w]) =
                make_double2(wn, __longlong_as_double((long long)(unsigned)(wp & 0x7fffffff) | ((long long)(unsigned)ws << 32)));
        }
    }
    QB_STAMP(2)
    QC_BARRIER();
    QB_STAMP(3)
    // ---- the step's pivot among the candidates (all loads first, then a branch-free tree)
    double maxval;
    int jm, sp;
    {
        double cn[NW];
        int cp[NW], cs[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const double2 cq = *reinterpret_cast<const double2 *>(sm.wc[par][q]);
            const long long pc = __double_as_longlong(cq.y);
            cn[q] = cq.x;
            cp[q] = (int)(pc & 0x7fffffff);
            cs[q] = (int)(pc >> 32);
        }
#pragma unroll
        for (int st = 1; st < NW; st *= 2) {
#pragma unroll
            for (int q = 0; q + st < NW; q += 2 * st) qb_merge(cn[q], cp[q], cs[q], cn[q + st], cp[q + st], cs[q + st]);
        }
        maxval = cn[0]; jm = cp[0]; sp = cs[0];
    }
    QB_STAMP(4)
    const bool none = sp < 0;  // nothing live (only with non-finite norms): no reflector, no swap
    maxval = none ? 0.0 : maxval;
    jm = none ? j : jm;
    const double *cb = sm.colbuf[par][none ? 0 : ((sp % NG) >> 3)];
    const double xj = cb[(t & 7) * VS + (t >> 3)];
    double2 cvv[KR / 2];
    {
        const double2 *vq = reinterpret_cast<const double2 *>(cb + rg * VS);
#pragma unroll
        for (int k = KB0; k < KR; k += 2) cvv[k >> 1] = vq[k >> 1];
    }
    const double cvo = tid < G::M ? cb[(tid & 7) * VS + (tid >> 3)] : 0.0;
    // ---- reflector scalars (UDT.jl:133-148; branch-free: a zero column keeps tau = 0 and the column as it is)
    const bool nz = maxval != 0.0;
    double rootn, rrootn;
    qb_sqrt_rsqrt(nz ? maxval : 1.0, rootn, rrootn);
    const double nu = nz ? copysign(rootn, xj) : 1.0;
    const double xi = nz ? xj + nu : 1.0;
    const double tj = nz ? __builtin_fma(fabs(xj), rrootn, 1.0) : 0.0;  // xi / nu
    const double rcp = qb_rcp(xi);
    // swap positions j <-> jm (UDT.jl:219-231): every thread tracks its own slots
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int s = g + NG * c;
        mypos[c] = (s == sp) ? j : ((mypos[c] == j) ? jm : mypos[c]);
    }
    if (tid == 0) tau[j] = tj;
    if (tid < G::M) {  // rows >= J0 of output column j
        const int r = J0 + tid;
        Wo[r + 256l * j] = (nz & (r == j)) ? -nu : ((nz & (r > j)) ? cvo * rcp : cvo);
    }
    QB_STAMP(5)
    // ---- H_j on my columns with the unscaled vector u = xi v (u_j = xi), fresh norms of rows > j
    double u[KR];
#pragma unroll
    for (int k = KB0; k < KR; k += 2) {
        if (k < KB0 + 4) {
            const int r0 = J0 + rg + 8 * k, r1 = r0 + 8;
            u[k] = r0 > j ? cvv[k >> 1].x : (r0 == j ? xi : 0.0);
            u[k + 1] = r1 > j ? cvv[k >> 1].y : (r1 == j ? xi : 0.0);
        } else {
            u[k] = cvv[k >> 1].x;
            u[k + 1] = cvv[k >> 1].y;
        }
    }
    const double trr = tj * rcp;
    double wv[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = KB0; k < KR; k += 2) {
            d0 = __builtin_fma(u[k], x[c][k], d0);
            d1 = __builtin_fma(u[k + 1], x[c][k + 1], d1);
        }
        wv[c] = d0 + d1;
    }
    QB_STAMP(6)
#pragma unroll
    for (int c = 0; c < CPT; ++c) wv[c] = (sum8(wv[c]) * rcp) * trr;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        double n0 = 0.0, n1 = 0.0;
#pragma unroll
        for (int k = KB0; k < KR; k += 2) {
            const double y0 = __builtin_fma(-u[k], wv[c], x[c][k]), y1 = __builtin_fma(-u[k + 1], wv[c], x[c][k + 1]);
            x[c][k] = y0;
            x[c][k + 1] = y1;
#ifndef QR_NONORM_EXPERIMENT
            if (k < KB0 + 4) {
                n0 += (J0 + rg + 8 * k > j ? 1.0 : 0.0) * (y0 * y0);
                n1 += (J0 + rg + 8 * (k + 1) > j ? 1.0 : 0.0) * (y1 * y1);
            } else {
                n0 = __builtin_fma(y0, y0, n0);
                n1 = __builtin_fma(y1, y1, n1);
            }
#endif
        }
#ifdef QR_NONORM_EXPERIMENT
        nrm[c] = nrm[c] * 0.999 + wv[c] * 1e-300;
#else
        nrm[c] = n0 + n1;
#endif
    }
#ifndef QR_NONORM_EXPERIMENT
#pragma unroll
    for (int c = 0; c < CPT; ++c) nrm[c] = sum8(nrm[c]);
#endif
    QB_STAMP(7)
}

// ---- the last 64 steps on ONE wave, one column per lane (qr_tail_kernel<128, 4> hands over at j = 192) -----------
// Lane l holds rows 192..255 of the column that sits at position 192 + l at the hand-over: a reflector is applied with
// no cross-lane reduction at all, the pivot search is one wave-wide DPP reduction, and the pivot column reaches the
// other lanes as LDS broadcast reads - no workgroup barrier, no second wave.  Pivoting stays logical.
constexpr int QF_LD = 65;
struct QfShared {
    double Mx[64 * QF_LD];                         // hand-over: row r - 192 of the column at position 192 + p at [r][p]
    __attribute__((aligned(16))) double ub[64];    // the step's reflector (unscaled, u_j = xi), rows 192..255
    int cidp[64];                                  // original column at position 192 + p (hand-over order)
};
// R8: steps t = 8 R8 .. 8 R8 + 7 (relative to 192); rows below 8 R8 are finished in every live column
template <int R8>
__device__ __forceinline__ void qf_step(int t, double (&x)[64], double &nrm, int &mypos, double &s_nu, double &s_rcp,
                                        double &s_tau, int &s_nz, QfShared &fs)
{
    constexpr int R0 = 8 * R8, NR = 64 - R0;
    const int lane = threadIdx.x & 63, j = 192 + t;
    // ---- pivot: largest norm among the live columns, then smallest position (UDT.jl:151-168)
    const double bn = (mypos >= j) ? nrm : -1.0;
    const double wn = wave_max_f64<1>(bn);
    const bool eq = (bn == wn) & (bn >= 0.0);
    unsigned long long m = __ballot(eq);
    if (__popcll(m) > 1) {
        const unsigned wp = wave_min_u32<1>(eq ? (unsigned)mypos : 0xffffffffu);
        m = __ballot(eq & ((unsigned)mypos == wp));
    }
    double maxval = wn;
    if (m == 0) {  // nothing live (non-finite norms only): the column at position j, no reflector
        m = __ballot(mypos == j);
        maxval = 0.0;
    }
    const int pl = __ffsll((long long)m) - 1;
    const int jm = __builtin_amdgcn_readlane(mypos, pl);
    // element j of every lane's own column, then the pivot lane's one as a wave-uniform value
    double xown = x[R0];
#pragma unroll
    for (int i = 1; i < 8; ++i) xown = (t - R0 == i) ? x[R0 + i] : xown;
    const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xown), pl),
                                       __builtin_amdgcn_readlane(__double2loint(xown), pl));
    // ---- reflector scalars (UDT.jl:133-148)
    const bool nz = maxval != 0.0;
    double rootn, rrootn;
    qb_sqrt_rsqrt(nz ? maxval : 1.0, rootn, rrootn);
    const double nu = nz ? copysign(rootn, xj) : 1.0;
    const double xi = nz ? xj + nu : 1.0;
    const double tj = nz ? __builtin_fma(fabs(xj), rrootn, 1.0) : 0.0;
    const double rcp = qb_rcp(xi);
    const double trr = tj * rcp;
    // ---- the pivot lane publishes u (rows <= j masked, u_j = xi) and keeps what its output column needs
    if (lane == pl) {
        double2 *ub2 = reinterpret_cast<double2 *>(fs.ub + R0);
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            double a = x[R0 + i], b = x[R0 + i + 1];
            if (i < 8) {
                a = (R0 + i > t) ? a : ((R0 + i == t) ? xi : 0.0);
                b = (R0 + i + 1 > t) ? b : ((R0 + i + 1 == t) ? xi : 0.0);
            }
            ub2[i >> 1] = make_double2(a, b);
        }
        s_nu = nu; s_rcp = rcp; s_tau = tj; s_nz = nz ? 1 : 0;
    }
    // swap positions j <-> jm (UDT.jl:219-231)
    mypos = (lane == pl) ? j : ((mypos == j) ? jm : mypos);
    // ---- H_j on the live columns (one per lane): dot, update, fresh norm of rows > j; two passes over u in LDS
    if (mypos > j) {
        const double2 *ub2 = reinterpret_cast<const double2 *>(fs.ub + R0);
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            const double2 uu = ub2[i >> 1];
            d0 = __builtin_fma(uu.x, x[R0 + i], d0);
            d1 = __builtin_fma(uu.y, x[R0 + i + 1], d1);
        }
        const double wv = ((d0 + d1) * rcp) * trr;
        asm volatile("" ::: "memory");  // second pass re-reads u from LDS instead of keeping 64 doubles in registers
        double n0 = 0.0, n1 = 0.0;
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            const double2 uu = ub2[i >> 1];
            const double y0 = __builtin_fma(-uu.x, wv, x[R0 + i]), y1 = __builtin_fma(-uu.y, wv, x[R0 + i + 1]);
            x[R0 + i] = y0;
            x[R0 + i + 1] = y1;
#ifndef QR_NONORM_EXPERIMENT
            if (i < 8) {
                n0 += (R0 + i > t ? 1.0 : 0.0) * (y0 * y0);
                n1 += (R0 + i + 1 > t ? 1.0 : 0.0) * (y1 * y1);
            } else {
                n0 = __builtin_fma(y0, y0, n0);
                n1 = __builtin_fma(y1, y1, n1);
            }
#endif
        }
#ifdef QR_NONORM_EXPERIMENT
        nrm = nrm * 0.999 + wv * 1e-300;
#else
        nrm = n0 + n1;
#endif
    }
}

template <int J0, int NW>
__global__ __launch_bounds__(64 * NW) void qr_tail_kernel(int n_units, double *__restrict__ Xall, long strideX,
                                                         double *__restrict__ Wall, long strideW,
                                                         double *__restrict__ tauall, int *__restrict__ pivall,
                                                         const int *fb, int epoch_i)
{
    using G = QbGeo<J0, NW>;
    constexpr int CPT = G::CPT, KR = G::KR, NG = 8 * NW;
    __shared__ QbShared<J0, NW> sm;
    if (fb[0] == epoch_i) return;  // the cooperative phase timed out: the guarded kernel behind redoes everything
    const int unit = blockIdx.x;
    double *__restrict__ X = Xall + (long)unit * strideX;
    double *__restrict__ Wo = Wall + (long)unit * strideW;
    double *__restrict__ tau = tauall + (long)unit * 256;
    int *__restrict__ piv = pivall + (long)unit * 256;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = lane & 7, cg = lane >> 3, g = 8 * w + cg;
    if (tid < G::M) sm.colid[tid] = piv[J0 + tid];
    __syncthreads();
    double x[CPT][KR], nrm[CPT];
    int mypos[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int s = g + NG * c;
        const double *__restrict__ Xc = X + 256l * sm.colid[s] + J0 + rg;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            x[c][k] = Xc[8 * k];
            a += x[c][k] * x[c][k];
        }
        nrm[c] = sum8(a);
        mypos[c] = J0 + s;
    }
    // FIN: the last 64 steps run on wave 0 alone, one column per lane (qf_step)
    constexpr bool FIN = (J0 >= 96 && NW <= 5);
    constexpr int NREG = FIN ? G::REGIONS - 2 : G::REGIONS;
#define QB_REGION(REG)                                                                     \
    if ((REG) < NREG)                                                                      \
        for (int t = 32 * (REG); t < 32 * (REG) + 32; ++t)                                 \
            qb_step<J0, NW, ((REG) < G::REGIONS ? 4 * (REG) : 0)>(t, x, nrm, mypos, sm, Wo, tau);
    QB_REGION(0) QB_REGION(1) QB_REGION(2) QB_REGION(3) QB_REGION(4) QB_REGION(5)
#undef QB_REGION
    static_assert(G::REGIONS <= 6, "qr_tail_kernel: at most six regions");
    if (FIN) {
        __shared__ QfShared fs;
        constexpr int KH = (192 - J0) / 8;  // row blocks below row 192
        // hand-over: rows >= 192 of the live columns to LDS by position, their finished rows J0..191 to X
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (mypos[c] >= 192) {
                const int p = mypos[c] - 192, cid = sm.colid[g + NG * c];
#pragma unroll
                for (int k = KH; k < KR; ++k) fs.Mx[(rg + 8 * (k - KH)) * QF_LD + p] = x[c][k];
#pragma unroll
                for (int k = 0; k < KH; ++k) X[256l * cid + J0 + rg + 8 * k] = x[c][k];
                if (rg == 0) fs.cidp[p] = cid;
            } else if (rg == 0) {
                sm.posmap[mypos[c] - J0] = sm.colid[g + NG * c];
            }
        }
        __syncthreads();
        if (w == 0) {
            double xc[64];
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int r = 0; r < 64; r += 2) {
                xc[r] = fs.Mx[r * QF_LD + lane];
                xc[r + 1] = fs.Mx[(r + 1) * QF_LD + lane];
                a0 = __builtin_fma(xc[r], xc[r], a0);
                a1 = __builtin_fma(xc[r + 1], xc[r + 1], a1);
            }
            double cn = a0 + a1, s_nu = 1.0, s_rcp = 1.0, s_tau = 0.0;
            int cpos = 192 + lane, s_nz = 0;
            const int cid = fs.cidp[lane];
#define QF_REGION(R) \
    for (int t = 8 * (R); t < 8 * (R) + 8; ++t) qf_step<(R)>(t, xc, cn, cpos, s_nu, s_rcp, s_tau, s_nz, fs);
            QF_REGION(0) QF_REGION(1) QF_REGION(2) QF_REGION(3) QF_REGION(4) QF_REGION(5) QF_REGION(6) QF_REGION(7)
#undef QF_REGION
            // every lane was the pivot exactly once: its column goes to its final position
            double *__restrict__ Wc = Wo + 256l * cpos + 192;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const int R = 192 + r;
                Wc[r] = (R < cpos || !s_nz) ? xc[r] : ((R == cpos) ? -s_nu : xc[r] * s_rcp);
            }
            tau[cpos] = s_tau;
            sm.posmap[cpos - J0] = cid;
        }
        __syncthreads();
        if (tid < G::M) piv[J0 + tid] = sm.posmap[tid];
        // finished rows of the trailing columns: rows < J0 of all of them, rows J0..191 of the last 64
        for (int i = tid; i < J0 * G::M; i += G::THREADS) {
            const int r = i % J0, p = i / J0;
            Wo[r + 256l * (J0 + p)] = X[r + 256l * sm.posmap[p]];
        }
        for (int i = tid; i < (192 - J0) * 64; i += G::THREADS) {
            const int r = J0 + i % (192 - J0), p = 192 - J0 + i / (192 - J0);
            Wo[r + 256l * (J0 + p)] = X[r + 256l * sm.posmap[p]];
        }
        return;
    }
    // final pivot vector and the finished rows < J0 of the trailing columns
    if (rg == 0) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) sm.posmap[mypos[c] - J0] = sm.colid[g + NG * c];
    }
    __syncthreads();
    if (tid < G::M) piv[J0 + tid] = sm.posmap[tid];
    for (int i = tid; i < J0 * G::M; i += G::THREADS) {
        const int r = i % J0, p = i / J0;
        Wo[r + 256l * (J0 + p)] = X[r + 256l * sm.posmap[p]];
    }
}

#ifdef QB_STAMPS
extern "C" int dqmc_debug_qb_stamps(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qb_stamp_ptr), &devptr, sizeof(void *));
}
#endif

// blocks of qr_coop_kernel that one CU holds at a time (occupancy API, capped at the 2 that its registers admit)
int qr_coop_blocks_per_cu()
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qr_coop_kernel, 256, 0) != hipSuccess) return 1;
    return nb < 1 ? 1 : (nb > 2 ? 2 : nb);
}

static hipError_t launch_qr_single(int n, int n_units, double *A, long strideA, double *tau, int *pivot,
                                   const double *src, long strideSrc, int *guard, int guard_val, hipStream_t s,
                                   double *X = nullptr, long strideX = 0, const int *never = nullptr)
{
    const bool no_tile = kernel_switches().qr_stream;
    if (n > 128 && n <= 256 && !no_tile) {
        const size_t lds_t = (64 * QT_LSTRIDE + 3 * 256 + 8 * QT_VS + 8 * 256) * sizeof(double) + 256 * sizeof(int);
        int dev = 0;
        (void)hipGetDevice(&dev);
        static unsigned attr_mask = 0;  // per device
        if (!(attr_mask & (1u << dev))) {
            (void)hipFuncSetAttribute((const void *)qr_tile256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_t);
            (void)hipFuncSetAttribute((const void *)qr_tile256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_t);
            attr_mask |= 1u << dev;
        }
        // X given (primary use with more matrices than the cooperative kernel can hold, n == 256): the tile kernel does
        // the first 128 steps, qr_tail_kernel the rest; `never` is a word that never equals -1 (the tail's exit test)
        const bool two_phase = X && never && n == 256 && !guard;
        const int nsteps = two_phase ? 128 : n;
        if (n == 256 && !kernel_switches().qr_tile_bounds)
            hipLaunchKernelGGL(qr_tile256_kernel<true>, dim3(n_units), dim3(QT_THREADS), lds_t, s, n, nsteps, A, strideA, tau,
                               pivot, src, strideSrc, guard, guard_val, two_phase ? X : nullptr, strideX);
        else
            hipLaunchKernelGGL(qr_tile256_kernel<false>, dim3(n_units), dim3(QT_THREADS), lds_t, s, n, nsteps, A, strideA, tau,
                               pivot, src, strideSrc, guard, guard_val, two_phase ? X : nullptr, strideX);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || !two_phase) return e;
        hipLaunchKernelGGL((qr_tail_kernel<128, 4>), dim3(n_units), dim3(256), 0, s, n_units, X, strideX, A, strideA, tau, pivot,
                           never, -1);
        return hipGetLastError();
    }
    // n > 256: the panel (dlaqps-style) kernel, unless it is switched off or its LDS does not fit
    if (n > 256 && n <= 768 && !no_tile && !guard && !src && !kernel_switches().qr_nopanel) {
        const size_t lds_p = ((size_t)(QP_NB + 4) * n + 2 * QP_NB + QP_WAVES + 2) * sizeof(double) + (size_t)(n + 2 + QP_WAVES) * sizeof(int);
        int dev = 0;
        (void)hipGetDevice(&dev);
        static unsigned pmask = 0;
        if (!(pmask & (1u << dev))) {
            (void)hipFuncSetAttribute((const void *)qr_panel_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipFuncSetAttribute((const void *)qr_panel_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            pmask |= 1u << dev;
        }
        const double thr = kernel_switches().qp_thr;  // recompute threshold (A/B)
        if (n <= 576) hipLaunchKernelGGL((qr_panel_kernel<9>), dim3(n_units), dim3(QP_THREADS), lds_p, s, n, A, strideA, tau, pivot, thr);
        else hipLaunchKernelGGL((qr_panel_kernel<12>), dim3(n_units), dim3(QP_THREADS), lds_p, s, n, A, strideA, tau, pivot, thr);
        return hipGetLastError();
    }
    const size_t lds = 2 * 1024 * sizeof(double);
    dim3 grid(n_units), block(QR_THREADS);
#define QR_LAUNCH(Q, UC) \
    hipLaunchKernelGGL((qr_pivot_kernel<Q, UC>), grid, block, lds, s, n, A, strideA, tau, pivot, src, strideSrc, guard, guard_val)
    if (n <= 64) QR_LAUNCH(1, 4);
    else if (n <= 128) QR_LAUNCH(2, 4);
    else if (n <= 256) QR_LAUNCH(4, 4);
    else if (n <= 576) QR_LAUNCH(9, 2);
    else QR_LAUNCH(16, 1);
#undef QR_LAUNCH
    return hipGetLastError();
}

// *factored: where the factored matrix is (W for the cooperative kernel, which leaves A untouched; else A, in place)
hipError_t launch_qr_pivot(int n, int n_units, double *A, long strideA, double *tau, int *pivot, QrCoopWorkspace *ws,
                           double *W, long strideW, const double **factored, hipStream_t s, double *X, long strideX)
{
    if (n > 1024) return hipErrorInvalidValue;
    *factored = A;
    // cooperative kernel: all 8 workgroups of every unit must be co-resident (they wait for each other); the grid is
    // admitted only if it fits the occupancy the runtime reports for this kernel.  Should another stream hold CUs
    // for longer than the bounded spins allow, the launch gives up and the guarded kernel behind it takes over.
    if (ws && ws->mailbox && W && n <= 256 && !ws->no_coop) {
        const int groups = (n_units + 7) / 8;
        const int blocks = groups * 8 * QC_PARTS;
        if (blocks <= ws->max_blocks) {
            ws->epoch += 1;
            const int force_sc1 = ws->force_sc1, force_to = ws->force_timeout;
            // n == 256: the first steps cooperatively, the rest on one CU per matrix (qr_tail_kernel)
            const int tail_j0 = ws->tail_j0;  // DQMC_QR_TAIL at handle creation; default 128 x 128 tail
            const bool two_phase = n == 256 && X && (tail_j0 == 64 || tail_j0 == 96 || tail_j0 == 128);
            hipLaunchKernelGGL(qr_coop_kernel, dim3(blocks), dim3(256), 0, s, n, n_units, A, strideA, tau, pivot,
                               ws->mailbox, ws->epoch, ws->fb, force_sc1, W, strideW, force_to, two_phase ? tail_j0 : n,
                               X, strideX);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            if (two_phase) {
                const int ep = (int)(ws->epoch & 0x7fffffffull);
                if (tail_j0 == 64)
                    hipLaunchKernelGGL((qr_tail_kernel<64, 8>), dim3(n_units), dim3(512), 0, s, n_units, X, strideX, W, strideW,
                                       tau, pivot, ws->fb, ep);
                else if (tail_j0 == 96)
                    hipLaunchKernelGGL((qr_tail_kernel<96, 5>), dim3(n_units), dim3(320), 0, s, n_units, X, strideX, W, strideW,
                                       tau, pivot, ws->fb, ep);
                else
                    hipLaunchKernelGGL((qr_tail_kernel<128, 4>), dim3(n_units), dim3(256), 0, s, n_units, X, strideX, W, strideW,
                                       tau, pivot, ws->fb, ep);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            *factored = W;
            return launch_qr_single(n, n_units, W, strideW, tau, pivot, A, strideA, ws->fb,
                                    (int)(ws->epoch & 0x7fffffffull), s);
        }
    }
    const bool tail_ok = ws && ws->fb && ws->tail_j0 == 128;  // (fb[0] holds launch epochs >= 0, never -1)
    return launch_qr_single(n, n_units, A, strideA, tau, pivot, nullptr, 0, nullptr, 0, s, tail_ok ? X : nullptr, strideX,
                            tail_ok ? ws->fb : nullptr);
}

// D = |diag R| (UDT.jl:268-272) for every unit
__global__ __launch_bounds__(256) void udt_diag_kernel(int n, int n_units, const double *__restrict__ Fall, long strideF,
                                                      double *__restrict__ Dall, long strideD)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n_units * n) return;
    const int unit = (int)(idx / n), i = (int)(idx - (long)unit * n);
    Dall[(long)unit * strideD + i] = fabs(Fall[(long)unit * strideF + i + (long)n * i]);
}

// D, V and T from the factored matrix (UDT.jl:268-306).  One workgroup per unit.
// workgroups per matrix (interleaved columns): a pure copy / scale pass over 3 n^2 doubles per matrix, bandwidth-bound;
// 32 matrices x 8 workgroups left the CUs at one workgroup each (20.6 us per batch at n = 256, 2.4 TB/s)
static inline int uf_split(int n) { return n >= 128 ? 32 : 8; }
__global__ __launch_bounds__(256) void udt_finish_kernel(int n, double *Aall, long strideA,
                                                        const double *Fall, long strideF,
                                                        const int *__restrict__ pivall,
                                                        double *__restrict__ Dall, long strideD,
                                                        double *__restrict__ Vall, long strideV,
                                                        double *__restrict__ Tall, long strideT, int apply_pivot,
                                                        int d_ready, int UF_SPLIT)
{
    extern __shared__ __attribute__((aligned(16))) double dinv[];  // 1/D
    const int unit = blockIdx.x / UF_SPLIT, slab = blockIdx.x % UF_SPLIT;
    double *A = Aall + (long)unit * strideA;
    const double *F = Fall + (long)unit * strideF;  // the factored matrix (may be A itself: no __restrict__)
    double *__restrict__ D = Dall + (long)unit * strideD;
    const int *__restrict__ piv = pivall + (long)unit * n;
    double *__restrict__ V = Vall ? Vall + (long)unit * strideV : nullptr;
    double *__restrict__ T = Tall ? Tall + (long)unit * strideT : nullptr;
    // d_ready: D was taken from the diagonal by udt_diag_kernel in a launch of its own.  That form is mandatory when
    // this kernel rescales the factored matrix IN PLACE (F == A, apply_pivot == 0): the diagonal entry (i, i) is
    // rewritten to +-1 by the workgroup that owns column i, and a workgroup of the same matrix that starts later
    // would otherwise read |+-1| instead of D[i] (the wrong Green's function of gpurun_out/r02_t3.log)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = d_ready ? D[i] : fabs(F[i + (long)n * i]);
        if (slab == 0 && !d_ready) D[i] = d;
        dinv[i] = 1.0 / d;
    }
    __syncthreads();
    const int rpt = (n + 63) / 64;  // rows handled per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = slab * nw + wave; c < n; c += nw * UF_SPLIT) {
        const int pc = apply_pivot ? piv[c] : c;
        for (int q = 0; q < rpt; ++q) {
            const int r = lane + 64 * q;
            if (r >= n) break;
            const double a = F[r + (long)n * c];
            if (V) V[r + (long)n * c] = r > c ? a : (r == c ? 1.0 : 0.0);
            if (apply_pivot) T[r + (long)n * pc] = r <= c ? dinv[r] * a : 0.0;
            else if (r <= c) A[r + (long)n * c] = dinv[r] * a;
        }
    }
}

hipError_t launch_udt_finish(int n, int n_units, double *A, long strideA, const double *F, long strideF,
                             const int *pivot, double *D, long strideD, double *V, long strideV, double *Tout,
                             long strideT, int apply_pivot, hipStream_t s)
{
    // in-place rescaling (Val(false) behind one of the in-place factorisations): D in a launch of its own, see the kernel
    const int d_ready = (!apply_pivot && F == A) ? 1 : 0;
    if (d_ready) {
        hipLaunchKernelGGL(udt_diag_kernel, dim3((n_units * n + 255) / 256), dim3(256), 0, s, n, n_units, F, strideF, D,
                           strideD);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int split = uf_split(n);
    hipLaunchKernelGGL(udt_finish_kernel, dim3(n_units * split), dim3(256), n * sizeof(double), s, n, A, strideA,
                       F, strideF, pivot, D, strideD, V, strideV, Tout, strideT, apply_pivot, d_ready, split);
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

// ---------------------------------------------------------------------------
// MFMA-blocked variant for n <= 256.  X T = O with T upper triangular is solved by column blocks
// of 16: the contribution of all previous blocks is one 32 x 16 x j0 product on
// v_mfma_f64_16x16x4_f64 (X slab from LDS, the T panel staged to LDS), followed by a 16-step
// substitution inside the block (one lane per row, the block's solved values in registers).
// One workgroup (2 waves) owns a 32-row slab; 8 slabs x units fill the chip.
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int TM_XS = 40;    // LDS stride of an X column (32 rows + pad: conflict-free MFMA operand reads)
constexpr int TM_TS = 258;   // LDS stride of a T panel column (2*TS = 4 mod 64 dwords: conflict-free)

// Inverses of the 16 x 16 diagonal blocks of T (upper triangular; reciprocal diagonal either 1/T[c,c] or
// dmul[c]), one workgroup of 16 threads per (block, unit): thread j builds column j of W = B^-1 by back
// substitution.  With them the in-block solve of the main kernel is one small MFMA product instead of a
// 16-step substitution on 16 lanes.  Rows/columns beyond n are the identity.
__global__ __launch_bounds__(16) void trsm_diag_inv_kernel(int n, const double *__restrict__ Tall, long sT,
                                                          const double *__restrict__ dmulall, long sV,
                                                          double *__restrict__ Wall, int nblk)
{
    // (n, nblk: the whole matrix; every 16 x 16 diagonal block of it is inverted, also for the panelled solve)
    __shared__ double B[16][17], rdg[16];
    const int J = blockIdx.x, unit = blockIdx.y, j = threadIdx.x, j0 = J << 4;
    const double *__restrict__ T = Tall + (long)unit * sT;
    const double *__restrict__ dmul = dmulall ? dmulall + (long)unit * sV : nullptr;
    double *__restrict__ W = Wall + ((long)unit * nblk + J) * 256;
    const int col = j0 + j;
    {   // sixteen requests in flight, then the bounds as selects (as a conditional load each one waits for the one before)
        const int cc = min(col, n - 1);
        double bv[16], dv;
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = T[min(j0 + i, n - 1) + (long)n * cc];
        dv = dmul ? dmul[cc] : T[cc + (long)n * cc];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) B[i][j] = (col < n && j0 + i < n && i < j) ? bv[i] : 0.0;
        rdg[j] = col < n ? (dmul ? dv : 1.0 / dv) : 1.0;
    }
    __syncthreads();
    double wcol[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        double sum = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int kk = i + 1; kk < 16; ++kk) sum -= B[i][kk] * ((kk <= j) ? wcol[kk] : 0.0);
        wcol[i] = (i <= j) ? sum * rdg[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) W[i + 16 * j] = wcol[i];  // column-major 16 x 16
}

// Panel form: solves the nc columns c0 .. c0 + nc - 1 (nc <= 256) of an nr-row problem against the diagonal block
// T[c0 : c0 + nc, c0 : c0 + nc] of an ld x ld triangle; the contribution of the columns before c0 must already have
// been subtracted (launch_trsm_right_upper does that with a GEMM for n > 256).  nr = nc = ld = n, c0 = 0 is the
// whole n <= 256 problem.
__global__ __launch_bounds__(128) void trsm_mfma_kernel(int nr, int n, int c0, int ld, int nblk_all,
                                                       const double *__restrict__ Aall, long sA,
                                                       const double *__restrict__ Tall, long sT,
                                                       const int *__restrict__ pivall,
                                                       const double *__restrict__ dmulall, long sV,
                                                       double *__restrict__ Oall, long sO, int slabs,
                                                       const double *__restrict__ Wall)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm;                   // [256][TM_XS]
    double *Tp = Xs + 256 * TM_XS;     // [16][TM_TS]  panel T[0 : j0+16, j0 : j0+16], column c at Tp + c*TM_TS
    double *Wl = Tp + 16 * TM_TS;      // [16][18] inverse of the current diagonal block, W[k][c] at Wl + c*18 + k
    const int unit = blockIdx.x / slabs, row0 = (blockIdx.x % slabs) * 32;
    const double *__restrict__ A = Aall + (long)unit * sA;
    const double *__restrict__ T = Tall + (long)unit * sT + (long)ld * c0 + c0;  // the panel's diagonal block
    const int *__restrict__ piv = pivall ? pivall + (long)unit * ld : nullptr;
    (void)dmulall; (void)sV;  // the diagonal enters through the inverted blocks
    double *__restrict__ O = Oall + (long)unit * sO;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;

    for (int idx = tid; idx < n * 32; idx += 128) {
        const int j = idx >> 5, r = idx & 31, row = row0 + r;
        const int pj = piv ? piv[c0 + j] : c0 + j;
        Xs[j * TM_XS + r] = row < nr ? A[row + (long)ld * pj] : 0.0;
    }
    const int nblk = (n + 15) >> 4;
    for (int idx = n * 32 + tid; idx < nblk * 16 * 32; idx += 128) Xs[(idx >> 5) * TM_XS + (idx & 31)] = 0.0;  // columns >= n
    const double *__restrict__ Wu = Wall + ((long)unit * nblk_all + (c0 >> 4)) * 256;
    // the T panel of block J+1 is requested from global memory (L2) while block J computes:
    // thread (c = tid >> 3) holds k = (tid & 7) + 8 i of column j0 + c in registers until the LDS panel is free
    double pre[32], pre_w[2] = {0.0, 0.0};
    auto fetch_panel = [&](int J) {
        // no per-lane predicates: entries below the diagonal are never used (the product reads rows
        // k < j0 <= col, the substitution rows j0 + c2 < col), indices are clamped into the matrix, and
        // the number of 8-row steps is wave-uniform
        const int j0 = J << 4, c = tid >> 3, col = min(j0 + c, n - 1);
        const int steps = J < nblk ? ((min(j0 + 16, n) + 15) >> 4) << 1 : 0;  // 8-row steps, rounded up to a pair
        const double *__restrict__ tc = T + (long)ld * col;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            if (i < steps) {  // one wave-uniform guard per pair
                pre[i] = tc[min((tid & 7) + 8 * i, n - 1)];
                pre[i + 1] = tc[min((tid & 7) + 8 * (i + 1), n - 1)];
            }
        }
        if (J < nblk) {
            pre_w[0] = Wu[(long)J * 256 + tid];
            pre_w[1] = Wu[(long)J * 256 + 128 + tid];
        }
    };
    fetch_panel(0);
    for (int J = 0; J < nblk; ++J) {
        const int j0 = J << 4;
        __syncthreads();  // previous block's panel no longer read; first pass: Xs complete
        {
            const int c = tid >> 3, steps = ((min(j0 + 16, n) + 15) >> 4) << 1;  // as fetched: wave-uniform
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                if (i < steps) {
                    Tp[c * TM_TS + (tid & 7) + 8 * i] = pre[i];
                    Tp[c * TM_TS + (tid & 7) + 8 * (i + 1)] = pre[i + 1];
                }
            }
            Wl[(tid >> 4) * 18 + (tid & 15)] = pre_w[0];
            Wl[((tid + 128) >> 4) * 18 + (tid & 15)] = pre_w[1];
        }
        __syncthreads();
        fetch_panel(J + 1);
        // acc[i = lq + 4 r][c = li] = sum_{k < j0} X[16 w + i, k] T[k, j0 + c]; two independent
        // accumulators (a dependent MFMA chain would expose the 64+ cycle MFMA latency per k-step)
        d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // j0 is a multiple of 16: steps of 16 k with all eight operand reads issued before the MFMAs
        const double *xa = Xs + lq * TM_XS + 16 * w + li, *tb = Tp + li * TM_TS + lq;
        for (int kk = 0; kk < j0; kk += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = xa[(kk + 4 * u) * TM_XS];
                b[u] = tb[kk + 4 * u];
            }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
        }
        const d4_t acc = acc0 + acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(j0 + li) * TM_XS + 16 * w + lq + 4 * r] -= acc[r];
        // in-block solve X_J = R_J W_J with the precomputed inverse of the diagonal block: four k-steps on the
        // wave's own 16 rows (LDS ops of one wave are ordered: the subtraction above is visible)
        {
            d4_t xacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double a = Xs[(j0 + 4 * u + lq) * TM_XS + 16 * w + li];
                const double b = Wl[li * 18 + 4 * u + lq];
                xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, xacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[(j0 + li) * TM_XS + 16 * w + lq + 4 * r] = xacc[r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * 32; idx += 128) {
        const int j = idx >> 5, r = idx & 31, row = row0 + r;
        if (row < nr) O[row + (long)ld * (c0 + j)] = Xs[j * TM_XS + r];
    }
}

// W[:, j] = A[:, pivot[j]] (the gather of rdivp!, general.jl:143-148) for the panelled solve
__global__ void trsm_gather_kernel(int n, const double *__restrict__ Aall, long sA, const int *__restrict__ pivall,
                                   double *__restrict__ Wall, long sW)
{
    const int unit = blockIdx.y;
    const double *__restrict__ A = Aall + (long)unit * sA;
    double *__restrict__ W = Wall + (long)unit * sW;
    const int *__restrict__ piv = pivall ? pivall + (long)unit * n : nullptr;
    const long nn = (long)n * n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nn; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        W[idx] = A[r + (long)n * (piv ? piv[c] : c)];
    }
}

// right-looking register-resident solve (trsm_rl.hip)
hipError_t launch_trsm_rl(int nr, int nc, int c0, int ld, int nblk_all, int n_units, const double *A, long sA,
                          const double *T, long sT, const int *pivot, double *Out, long sO, const double *winv,
                          hipStream_t s);

hipError_t launch_trsm_right_upper(int n, int n_units, const double *A, long sA, const double *T, long sT,
                                   const int *pivot, const double *dmul, long sV, double *Out, long sO,
                                   double *winv, hipStream_t s, double *scratch)
{
    const bool no_mfma = kernel_switches().trsm_simple;
    const size_t lds_m = (256 * TM_XS + 16 * TM_TS + 16 * 18) * sizeof(double);
    auto set_attr = [&]() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        static unsigned attr_mask = 0;  // per device
        if (!(attr_mask & (1u << dev))) {
            (void)hipFuncSetAttribute((const void *)trsm_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_m);
            attr_mask |= 1u << dev;
        }
    };
    if (n <= 256 && winv && !no_mfma) {
        const int slabs = (n + 31) / 32, nblk = (n + 15) / 16;
        hipLaunchKernelGGL(trsm_diag_inv_kernel, dim3(nblk, n_units), dim3(16), 0, s, n, T, sT, dmul, sV, winv, nblk);
        set_attr();
        // (in place is fine for both kernels: a workgroup reads all entries of its own 32 rows before it writes any)
        const bool left_looking = kernel_switches().trsm_ll;  // the slab-in-LDS kernel (A/B measurements)
        if (left_looking)
            hipLaunchKernelGGL(trsm_mfma_kernel, dim3(n_units * slabs), dim3(128), lds_m, s, n, n, 0, n, nblk, A, sA, T, sT,
                               pivot, dmul, sV, Out, sO, slabs, winv);
        else
            return launch_trsm_rl(n, n, 0, n, nblk, n_units, A, sA, T, sT, pivot, Out, sO, winv, s);
        return hipGetLastError();
    }
    if (n > 256 && winv && scratch && sA == sO && !no_mfma) {
        // Panels of up to 256 columns: gather (pivot) into scratch, then per panel  X_P = (A_P - X_<P T_<P,P) inv(T_PP):
        // the bracket with the MFMA GEMM, the triangle with the panel form of the kernel above.
        const int nblk = (n + 15) / 16, slabs = (n + 31) / 32;
        int pw = 256;
        for (int cand = 256; cand >= 128; cand -= 16)  // equal panels when a width between 128 and 256 divides n
            if (n % cand == 0) { pw = cand; break; }
        hipLaunchKernelGGL(trsm_diag_inv_kernel, dim3(nblk, n_units), dim3(16), 0, s, n, T, sT, dmul, sV, winv, nblk);
        int bx = (int)(((long)n * n + 255) / 256);
        if (bx > 128) bx = 128;
        hipLaunchKernelGGL(trsm_gather_kernel, dim3(bx, n_units), dim3(256), 0, s, n, A, sA, pivot, scratch, sO);
        set_attr();
        for (int c0 = 0; c0 < n; c0 += pw) {
            const int nc = n - c0 < pw ? n - c0 : pw;
            if (c0 > 0) {
                GemmArgs g{};
                g.M = n; g.N = nc; g.K = c0;
                g.n_units = n_units; g.nb = 1;
                g.A = mat(scratch, sO, n);
                g.B = mat(T + (long)n * c0, sT, n);
                g.C = scratch + (long)n * c0; g.strideC = sO; g.ldc = n;
                g.kscale = g.colscale = g.rowscale = g.adddiag = vs_none();
                g.alpha = -1.0; g.ident = 0.0; g.beta = 1;
                hipError_t e = launch_gemm(g, s);
                if (e != hipSuccess) return e;
            }
            hipError_t er = launch_trsm_rl(n, nc, c0, n, nblk, n_units, scratch, sO, T, sT, nullptr, scratch, sO, winv, s);
            if (er != hipSuccess) return er;
        }
        hipError_t e = hipMemcpyAsync(Out, scratch, sizeof(double) * (size_t)n_units * sO, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
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

static KernelSwitches g_switches;
const KernelSwitches &kernel_switches() { return g_switches; }
void refresh_kernel_switches()
{
    KernelSwitches k;
    k.qr_stream = getenv("DQMC_QR_STREAM") != nullptr;
    k.qr_tile_bounds = getenv("DQMC_QR_TILE_BOUNDS") != nullptr;
    k.qr_nopanel = getenv("DQMC_QR_NOPANEL") != nullptr;
    if (const char *e = getenv("DQMC_QP_THR")) k.qp_thr = atof(e);
    k.trsm_simple = getenv("DQMC_TRSM_SIMPLE") != nullptr;
    k.trsm_ll = getenv("DQMC_TRSM_LL") != nullptr;
    k.trsm_bounds = getenv("DQMC_TRSM_BOUNDS") != nullptr;
    k.flush_ncp2 = getenv("DQMC_FLUSH_NCP2") != nullptr;
    if (const char *e = getenv("DQMC_GEMM_STAGGER")) k.gemm_stagger = atoi(e);
    g_switches = k;
}

}  // namespace dqmc
