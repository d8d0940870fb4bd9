// trsm_rl.hip — rdivp! (src/linalg/general.jl:138-166) / the compact-WY triangle as a right-looking, register-resident
// blocked solve on v_mfma_f64_16x16x4_f64.  Compiled with a high unroll threshold: the block loop must be fully
// unrolled so that the tiles keep static register indices.
#include "kernels.h"
#include <type_traits>

namespace dqmc {

typedef double d4_t __attribute__((ext_vector_type(4)));

// Right-looking form of the same solve, register resident: a workgroup owns 32 rows of X; wave (rt, par) keeps the
// TRANSPOSED 16 x 16 tiles X'[block L][rows of row tile rt] of the column blocks L = par (mod 2) in MFMA accumulator
// layout for the whole solve.  Step J: the two owners of block J finish it, X_J' = W_J' R_J' (four MFMAs on their own
// registers), and leave it in LDS as a register image; after ONE barrier every wave subtracts T[J, L]' X_J' from its
// later tiles (four MFMAs each, B operand = that image, A operand = the T row panel staged in LDS one step ahead).
// The owners of block J + 1 update that tile first, so consecutive steps overlap.  Panel parameters as above.
constexpr int TR_TS = 18;  // LDS stride of a column of the T row panel (16 rows + pad)
__global__ __launch_bounds__(256) void trsm_rl_kernel(int nr, int n, int c0, int ld, int nblk_all,
                                                     const double *__restrict__ Aall, long sA,
                                                     const double *__restrict__ Tall, long sT,
                                                     const int *__restrict__ pivall, double *__restrict__ Oall, long sO,
                                                     int slabs, const double *__restrict__ Wall, int n_units)
{
    // Tl[J % 3]: T[16 J + k][16 L + c] at (16 L + c) * TR_TS + k.  Three buffers: step J + 1 is staged while slower
    // waves may still read the panel of step J - 1 (one barrier per step bounds the skew to one step).
    extern __shared__ __attribute__((aligned(16))) double trl_sm[];
    double (*Tl)[256 * TR_TS] = reinterpret_cast<double (*)[256 * TR_TS]>(trl_sm);
    double (*Xi)[2][4 * 64] = reinterpret_cast<double (*)[2][4 * 64]>(trl_sm + 3 * 256 * TR_TS);  // [J & 1][rt]: image of X_J'
    double (*Wi)[4 * 64] = reinterpret_cast<double (*)[4 * 64]>(trl_sm + 3 * 256 * TR_TS + 2 * 2 * 256);  // [J & 1]: W_J'
    // XCD-aware map (as in gemm_f64.hip): the slabs of one unit share an XCD, so its T is fetched into one L2 only
    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int unit = (seq / slabs) * 8 + xcd, row0 = (seq % slabs) * 32;
    if (unit >= n_units) return;
    const double *__restrict__ A = Aall + (long)unit * sA;
    const double *__restrict__ T = Tall + (long)unit * sT + (long)ld * c0 + c0;
    const int *__restrict__ piv = pivall ? pivall + (long)unit * ld : nullptr;
    double *__restrict__ O = Oall + (long)unit * sO;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, ci = lane & 15;
    const int rt = w & 1, par = w >> 1;
    const int nblk = (n + 15) >> 4;
    const double *__restrict__ Wu = Wall + ((long)unit * nblk_all + (c0 >> 4)) * 256;
    const int row = row0 + 16 * rt + ci;
    const int rowc = min(row, nr - 1);

    // my tiles: block L = 2 t + par; element [4 r + g][ci] = X[row][c0 + 16 L + 4 r + g]
    d4_t x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int L = 2 * t + par;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * L + 4 * r + g;
            const int cc = min(col, n - 1);
            const int pj = piv ? piv[c0 + cc] : c0 + cc;
            const double v = A[rowc + (long)ld * pj];
            x[t][r] = (L < nblk && col < n && row < nr) ? v : 0.0;
        }
    }
    // Staging of step J: the T row panel (rows 16 J .. +15 of the block, later columns) and W_J'.  Requested into
    // registers one step before it is written to LDS (two steps before it is read), so no step waits for L2:
    // thread -> row k = tid & 15 of the columns c = (tid >> 4) + 16 i: one instruction covers 4 columns x 128 bytes.
    double pv[16], pw = 0.0;
    auto request = [&](int J) {
        if (J >= nblk) return;
        const int k = min(16 * J + (tid & 15), n - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = min((tid >> 4) + 16 * i, n - 1);
            pv[i] = T[(long)ld * c + k];
        }
        // W_J (16 x 16, column-major W[k + 16 c]) as the A operand of X_J' = W_J' R_J':  A(i = ci, k = 4 q + g) = W[4 q + g][ci]
        const int q = tid >> 6, l = tid & 63;
        pw = Wu[(long)J * 256 + (4 * q + (l >> 4)) + 16 * (l & 15)];
    };
    auto deposit = [&](int J) {
        if (J >= nblk) return;
        double *tl = Tl[J % 3];
        const int k = tid & 15;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (tid >> 4) + 16 * i;
            const bool live = c < n && c >= 16 * (J + 1) && 16 * J + k < n;
            tl[c * TR_TS + k] = live ? pv[i] : 0.0;
        }
        Wi[J & 1][tid] = pw;
    };
    request(0);
    deposit(0);
    request(1);
    __syncthreads();
    // one step; J is a compile-time constant (tile indices must be static: the tiles live in registers)
    auto step = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        if (J >= nblk) return;
        constexpr int t_own = J >> 1;
        if ((J & 1) == par) {  // owner of block J (wave-uniform): finish it
            d4_t o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) o = __builtin_amdgcn_mfma_f64_16x16x4f64(Wi[J & 1][q * 64 + lane], x[t_own][q], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Xi[J & 1][rt][r * 64 + lane] = o[r];
                const int col = 16 * J + 4 * r + g;
                if (row < nr && col < n) O[row + (long)ld * (c0 + col)] = o[r];
            }
        }
        deposit(J + 1);
        request(J + 2);
        // X_J' and the staging of step J + 1 are visible; buffers of step J - 1 are free.  LDS-only barrier:
        // __syncthreads() would also drain the global loads just requested for step J + 2
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const double *xi = Xi[J & 1][rt];
        const double *tl = Tl[J % 3];
        const double b0 = xi[lane], b1 = xi[64 + lane], b2 = xi[128 + lane], b3 = xi[192 + lane];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int L = 2 * t + par;
            if (L <= J) continue;          // compile time
            if (L >= nblk) continue;       // wave-uniform
            const double *ta = tl + (16 * L + ci) * TR_TS + g;  // A(i = ci, k = 4 q + g) = T[16 J + 4 q + g][16 L + ci]
            x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ta[0], b0, x[t], 0, 0, 0);
            x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ta[4], b1, x[t], 0, 0, 0);
            x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ta[8], b2, x[t], 0, 0, 0);
            x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ta[12], b3, x[t], 0, 0, 0);
        }
    };
#define TRL_STEP(K) step(std::integral_constant<int, K>{})
    TRL_STEP(0); TRL_STEP(1); TRL_STEP(2); TRL_STEP(3); TRL_STEP(4); TRL_STEP(5); TRL_STEP(6); TRL_STEP(7);
    TRL_STEP(8); TRL_STEP(9); TRL_STEP(10); TRL_STEP(11); TRL_STEP(12); TRL_STEP(13); TRL_STEP(14); TRL_STEP(15);
#undef TRL_STEP
}

hipError_t launch_trsm_rl(int nr, int nc, int c0, int ld, int nblk_all, int n_units, const double *A, long sA,
                          const double *T, long sT, const int *pivot, double *Out, long sO, const double *winv,
                          hipStream_t s)
{
    if (nc > 256 || nc < 1) return hipErrorInvalidValue;
    const size_t lds_r = (3 * 256 * TR_TS + 2 * 2 * 256 + 2 * 256) * sizeof(double);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)trsm_rl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
        attr_mask |= 1u << dev;
    }
    const int slabs = (nr + 31) / 32;
    const int groups = (n_units + 7) / 8;
    hipLaunchKernelGGL(trsm_rl_kernel, dim3(groups * 8 * slabs), dim3(256), lds_r, s, nr, nc, c0, ld, nblk_all, A, sA, T, sT,
                       pivot, Out, sO, slabs, winv, n_units);
    return hipGetLastError();
}

}  // namespace dqmc
