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
#ifdef TR_STAMPS  // diagnostic build only (tools/tr_stamps.py): cycle stamps of workgroup 0
__device__ long long *tr_stamp_ptr = nullptr;
#define TR_STAMP(idx)                                                                                        \
    do {                                                                                                     \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tr_stamp_ptr)                                      \
            tr_stamp_ptr[(threadIdx.x >> 6) * 64 + (idx)] = (long long)__builtin_amdgcn_s_memtime();         \
    } while (0)
#else
#define TR_STAMP(idx) do { } while (0)
#endif
constexpr int TR_TS = 18;  // LDS stride of a column of the T row panel (16 rows + pad)
// FULLB: n == 256 columns (16 blocks) and whole 32-row slabs - no bounds anywhere, so a step is straight-line code (the
// scheduler can request a step's LDS operands together and issue its MFMAs back to back; with run-time bounds every tile
// sat behind a scalar branch and every pair of MFMAs waited for the LDS read issued just before it)
template <bool FULLB>
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
    double (*Wl)[4 * 64] = reinterpret_cast<double (*)[4 * 64]>(trl_sm + 3 * 256 * TR_TS + 2 * 2 * 256);  // [J]: W_J' (all 16)
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
    const int nblk = FULLB ? 16 : (n + 15) >> 4;
    const double *__restrict__ Wu = Wall + ((long)unit * nblk_all + (c0 >> 4)) * 256;
    const int row = row0 + 16 * rt + ci;
    const int rowc = min(row, nr - 1);

    TR_STAMP(0);
    // my tiles: block L = 2 t + par; element [4 r + g][ci] = X[row][c0 + 16 L + 4 r + g]
    // All 32 pivot entries first, then the 32 gathered columns, each load pinned in front of its bound test: with a plain
    // `ok ? A[..] : 0.0` the compiler sinks every load into an exec-mask branch of its own and waits for it at the merge
    // (s_waitcnt vmcnt(0)) - 32 trips to the L2 one after the other, 37 k cycles of a 120 k cycle kernel (stamps,
    // tools/tr_stamps.py).  Addresses are clamped, so the unconditional loads stay inside the matrix.
    d4_t x[8];
    int pjv[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = min(16 * (2 * t + par) + 4 * r + g, n - 1);
            pjv[t][r] = piv ? piv[c0 + cc] : c0 + cc;
        }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = A[(unsigned)(rowc + ld * pjv[t][r])];
    asm volatile("" ::: "memory");  // the 32 loads stay in front of the bound tests (see above), which are then plain selects
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int L = 2 * t + par;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * L + 4 * r + g;
            x[t][r] = (L < nblk && col < n && row < nr) ? x[t][r] : 0.0;
        }
    }
#ifdef TR_STAMPS
    TR_STAMP(40);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_STAMP(41);
#endif
    // Staging of step J: the T row panel (rows 16 J .. +15 of the block, later columns).  thread -> row k = tid & 15 of
    // the columns c = (tid >> 4) + 16 i: one instruction covers 4 columns x 128 bytes.  Register ring of THREE panels
    // (round 3: requested four steps before they are read, in LDS two steps before); the sixteen W_J' are staged in LDS
    // once, before the first step.
    constexpr int TR_RING = 3;
    double pv[TR_RING][16];
    auto request = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value, S = J % TR_RING;
        if (J >= nblk) return;
        const int k = FULLB ? 16 * J + (tid & 15) : min(16 * J + (tid & 15), n - 1);
#pragma unroll
        for (int i = (FULLB ? J + 1 : 0); i < 16; ++i) {  // (FULLB: only the columns of later blocks are ever read)
            const int c = FULLB ? (tid >> 4) + 16 * i : min((tid >> 4) + 16 * i, n - 1);
            pv[S][i] = T[(unsigned)(ld * c + k)];  // (32-bit offsets from the unit's scalar base: one VALU op per address)
        }
    };
    auto deposit = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value, S = J % TR_RING;
        if (J >= nblk) return;
        double *tl = Tl[J % 3];
        const int k = tid & 15;
#pragma unroll
        for (int i = (FULLB ? J + 1 : 0); i < 16; ++i) {
            const int c = (tid >> 4) + 16 * i;
            const bool live = FULLB || (c < n && c >= 16 * (J + 1) && 16 * J + k < n);
            tl[c * TR_TS + k] = live ? pv[S][i] : 0.0;
        }
    };
    // owner of block J (wave-uniform): X_J' = W_J' R_J' from its own registers -> LDS image + output
    auto finish = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value, t_own = J >> 1;
        d4_t o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) o = __builtin_amdgcn_mfma_f64_16x16x4f64(Wl[J][q * 64 + lane], x[t_own][q], o, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Xi[J & 1][rt][r * 64 + lane] = o[r];
            const int col = 16 * J + 4 * r + g;
            if (FULLB || (row < nr && col < n)) O[(unsigned)(row + ld * (c0 + col))] = o[r];
        }
    };
#define TRL_IC(K) std::integral_constant<int, (K)>{}
    request(TRL_IC(0));
    request(TRL_IC(1));
    request(TRL_IC(2));
    {   // all W_J (16 x 16, column-major W[k + 16 c]) as A operands of X_J' = W_J' R_J':  A(i = ci, k = 4 q + g) = W[4 q + g][ci]
        const int q = tid >> 6, l = tid & 63, src = (4 * q + (l >> 4)) + 16 * (l & 15);
        double wv[16];
#pragma unroll
        for (int J = 0; J < 16; ++J) wv[J] = Wu[(unsigned)((FULLB ? J : min(J, nblk - 1)) * 256 + src)];
#pragma unroll
        for (int J = 0; J < 16; ++J) Wl[J][tid] = wv[J];
    }
#ifdef TR_STAMPS
    TR_STAMP(42);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_STAMP(43);
#endif
    deposit(TRL_IC(0));
    TR_STAMP(44);
    __syncthreads();
    TR_STAMP(1);
    if (par == 0) finish(TRL_IC(0));
    deposit(TRL_IC(1));
    request(TRL_IC(3));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // One step (J is a compile-time constant: the tiles live in registers).  Look-ahead (round 3): the owner of block
    // J + 1 brings that tile up to date FIRST and finishes it right away - image in LDS, output on its way - and only
    // then updates its other tiles, so the next step's X is ready when everybody arrives at the barrier.  (Before, a
    // wave issued all its updates - up to 32 MFMAs - before the four MFMAs the next step was waiting for: 4 - 6 k cycles
    // per step against ~2.3 k of MFMA issue.)
    auto step = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        if (!FULLB && J >= nblk) return;
        const double *xi = Xi[J & 1][rt];
        const double *tl = Tl[J % 3];
        // (-X_J' once per step: negating the T operand cost one VALU op per MFMA)
        const double b0 = -xi[lane], b1 = -xi[64 + lane], b2 = -xi[128 + lane], b3 = -xi[192 + lane];
        // (the four MFMAs of a tile depend on each other: tiles are interleaved, k-quad by k-quad, so that consecutive
        // MFMAs are independent - issued tile by tile a wave spent ~190 cycles per MFMA instead of 64)
        auto body = [&](auto pc) {
            constexpr int PAR = decltype(pc)::value;
            constexpr bool own_next = J + 1 < 16 && ((J + 1) & 1) == PAR;
            constexpr int t_next = (J + 1) >> 1;  // tile index of block J + 1 in its owner
            if constexpr (own_next) {
                if (FULLB || J + 1 < nblk) {  // look-ahead: block J + 1 first, and finished right away
                    const double *ta = tl + (16 * (J + 1) + ci) * TR_TS + g;
                    x[t_next] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[0], b0, x[t_next], 0, 0, 0);
                    x[t_next] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[4], b1, x[t_next], 0, 0, 0);
                    x[t_next] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[8], b2, x[t_next], 0, 0, 0);
                    x[t_next] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[12], b3, x[t_next], 0, 0, 0);
                    finish(TRL_IC(J + 1 < 16 ? J + 1 : 15));
                }
            }
            constexpr int t_first = own_next ? t_next + 1 : (J + 2 - PAR) / 2;  // first tile with 2 t + PAR > J (+ not t_next)
            const double bq[4] = {b0, b1, b2, b3};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = t_first; t < 8; ++t) {
                    if (!FULLB && 2 * t + PAR >= nblk) continue;  // wave-uniform
                    const double *ta = tl + (16 * (2 * t + PAR) + ci) * TR_TS + g;  // A(i = ci, k = 4 q + g) = T[16 J + 4 q + g][16 L + ci]
                    x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[4 * q], bq[q], x[t], 0, 0, 0);
                }
        };
        if (par == 0) body(TRL_IC(0));
        else body(TRL_IC(1));
        deposit(TRL_IC(J + 2));
        request(TRL_IC(J + 4));
        // X_{J+1}' and the panel of step J + 2 are visible; buffers of step J are free.  LDS-only barrier:
        // __syncthreads() would also drain the global loads requested for the coming steps
        TR_STAMP(2 + 2 * J);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TR_STAMP(3 + 2 * J);
    };
#define TRL_STEP(K) step(std::integral_constant<int, K>{})
    TRL_STEP(0); TRL_STEP(1); TRL_STEP(2); TRL_STEP(3); TRL_STEP(4); TRL_STEP(5); TRL_STEP(6); TRL_STEP(7);
    TRL_STEP(8); TRL_STEP(9); TRL_STEP(10); TRL_STEP(11); TRL_STEP(12); TRL_STEP(13); TRL_STEP(14); TRL_STEP(15);
#undef TRL_STEP
#undef TRL_IC
    TR_STAMP(34);
}
#ifdef TR_STAMPS
extern "C" int dqmc_debug_tr_stamps(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr_stamp_ptr), &devptr, sizeof(void *));
}
#endif

hipError_t launch_trsm_rl(int nr, int nc, int c0, int ld, int nblk_all, int n_units, const double *A, long sA,
                          const double *T, long sT, const int *pivot, double *Out, long sO, const double *winv,
                          hipStream_t s)
{
    if (nc > 256 || nc < 1) return hipErrorInvalidValue;
    const size_t lds_r = (3 * 256 * TR_TS + 2 * 2 * 256 + 16 * 256) * sizeof(double);  // 148 KB
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)trsm_rl_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
        (void)hipFuncSetAttribute((const void *)trsm_rl_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
        attr_mask |= 1u << dev;
    }
    const int slabs = (nr + 31) / 32;
    const int groups = (n_units + 7) / 8;
    if (nc == 256 && nr % 32 == 0 && !kernel_switches().trsm_bounds)
        hipLaunchKernelGGL(trsm_rl_kernel<true>, dim3(groups * 8 * slabs), dim3(256), lds_r, s, nr, nc, c0, ld, nblk_all, A, sA, T,
                           sT, pivot, Out, sO, slabs, winv, n_units);
    else
        hipLaunchKernelGGL(trsm_rl_kernel<false>, dim3(groups * 8 * slabs), dim3(256), lds_r, s, nr, nc, c0, ld, nblk_all, A, sA,
                           T, sT, pivot, Out, sO, slabs, winv, n_units);
    return hipGetLastError();
}

}  // namespace dqmc
