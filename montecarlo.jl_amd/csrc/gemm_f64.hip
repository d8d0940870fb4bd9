// gemm_f64.hip — batched fp64 GEMM on v_mfma_f64_16x16x4_f64 with the DQMC diagonal
// scalings fused into operand load / epilogue.
//
// Stands in for the vmul! family of src/linalg/general.jl:7-56 and for every
// multiply_*slice_matrix* of src/flavors/DQMC/slice_matrices.jl:42-76: the slice
// matrix B_l = eT2*Diagonal(eV_l) is never materialised (slice_matrices.jl:23-39
// does); eV_l is rebuilt from the Int8 HS field while the tile is staged.
//
// Tiling: one 64x64 C tile per 256-thread workgroup (4 waves as 2x2, each wave
// 2x2 MFMA tiles of 16x16), BK = 16, LDS double buffered with register-staged
// prefetch.  The MFMA is issued "transposed" (A-operand <- opB, B-operand <- opA)
// so that the accumulator's lane index runs along m, the contiguous direction of
// column-major C: stores are 128-byte segments.
#include "kernels.h"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace dqmc {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16;
// LDS row stride (doubles).  2*LS = 32 (mod 64) dwords puts the k and k+1 rows that
// one 32-lane half reads with ds_read_b64 on disjoint bank halves.
constexpr int LS = 80;

// "direct" tile: the 64-long index (m or n) is contiguous in memory.
//   element(c, k) = p[c + ld*k]; thread -> k = tid/16, c = (tid%16)*4 .. +3
// "transposing" tile: k is contiguous in memory.
//   element(c, k) = p[k + ld*c]; thread -> c = tid/4, k = (tid%4)*4 .. +3
// FULL: the tile lies inside the matrix (M, N multiples of 64, K of 16): no predicates at all.  The predicated
// form compiles to an exec-mask branch cascade per load and to conservative s_waitcnt vmcnt(0) at the merges.
template <bool KCONTIG, bool FULL>
__device__ __forceinline__ void tile_load(const double *__restrict__ p, int ld, int c0, int cdim,
                                          int k0, int kdim, int tid, double r[4])
{
    if (FULL) {
        const double *q = KCONTIG ? p + (long)ld * (c0 + (tid >> 2)) + k0 + ((tid & 3) << 2)
                                  : p + (long)ld * (k0 + (tid >> 4)) + c0 + ((tid & 15) << 2);
        r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
        return;
    }
    if (!KCONTIG) {
        const int kk = k0 + (tid >> 4), cc = c0 + ((tid & 15) << 2);
        const double *q = p + (long)ld * kk + cc;
        if (kk < kdim && cc + 3 < cdim) {
            r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (kk < kdim && cc + i < cdim) ? q[i] : 0.0;
        }
    } else {
        const int cc = c0 + (tid >> 2), kk = k0 + ((tid & 3) << 2);
        const double *q = p + (long)ld * cc + kk;
        if (cc < cdim && kk + 3 < kdim) {
            r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (cc < cdim && kk + i < kdim) ? q[i] : 0.0;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void tile_store(double (*Xs)[LS], int tid, const double r[4])
{
    // columns of row k are rotated by 4 (k >> 2) (mod 64): the transposing store below then spreads the four lanes
    // that share a column over 16 distinct 8-byte slots (unrotated they hit one bank, rows 4 apart are 640 dwords
    // apart); the reads of compute() stay conflict free (a row's 16 lanes remain consecutive mod 64)
    if (!KCONTIG) {
        const int k = tid >> 4;
        double *d = &Xs[k][(((tid & 15) << 2) + ((k >> 2) << 2)) & 63];
        d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3];
    } else {
        const int c = tid >> 2, k = (tid & 3) << 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) Xs[k + i][(c + k) & 63] = r[i];
    }
}

// Barrier of the k-loop: only the LDS tiles are handed between waves there.  __syncthreads() would also wait for
// the global loads of the tile two steps ahead that were requested a moment earlier (s_waitcnt vmcnt(0)), i.e. put
// an L2 round trip into every k-step.
#define GEMM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// KSM: the k-scaling mode as a compile-time constant (0 none, 2 conf-derived, 1/3/4/5 the array modes; -1 = read it from the
// arguments: partial tiles only): the
// per-k-tile mode dispatch, its register copies and its divisions leave the loop for the two common cases
template <bool TA, bool TB, bool FULL, int KSM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, int tiles_m, int tiles_n, int gemm_stagger)
{
    __shared__ double As[2][BK][LS];
    __shared__ double Bs[2][BK][LS];

    // XCD-aware map: consecutive block ids round-robin over the 8 XCDs, so all
    // tiles of one unit are given ids with equal (id % 8) and share that XCD's L2.
    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int T = tiles_m * tiles_n;
    const int unit = (seq / T) * 8 + xcd;
    if (unit >= g.n_units) return;
    const int tile = seq % T;
    int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    int kbeg = 0, kend = g.K;
    if (g.tri == 1) {        // V'V: tiles below the diagonal are not needed, the sum starts at max(m0, n0) = n0
        if (m0 > n0) return;
        kbeg = n0;
    } else if (g.tri == 2) { // A V': V[n, k] = 0 for k > n
        kend = min(g.K, n0 + BN);
    }
    const int blk = unit % g.nb;

    const double *__restrict__ A = g.A.p + (long)unit * g.A.stride_unit + (long)blk * g.A.stride_blk;
    const double *__restrict__ B = g.B.p + (long)unit * g.B.stride_unit + (long)blk * g.B.stride_blk;
    double *__restrict__ C = g.C + (long)unit * g.strideC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int li = lane & 15, lq = lane >> 4;

    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nk = (kend - kbeg + BK - 1) / BK;
    // Register ring of two k-tiles: the loads of tile kt+2 are issued before the MFMAs of tile kt, so a tile
    // has two iterations to arrive (with 32 units the grid gives each CU only two workgroups, and one
    // iteration of MFMA work is shorter than an L2 round trip).
    double ra[2][4], rb[2][4];
    // The k-scaling (B_l = eT2 Diagonal(eV_l), ...) is applied when a tile goes to LDS, not when it is requested:
    // multiplying at load time would wait for the tile (and for the HS-field byte behind the factor) right away
    // and serialise the prefetch.  Until then the raw source of the factor sits in a register: the double of an
    // array-type VecSrc or the Int8 spin of the conf-derived one (TA: four consecutive k per thread, else one).
    constexpr int NKS = TA ? 4 : 1;
    double ksd[2][NKS];
    int ksc[2][NKS];
    const int ksmode = KSM >= 0 ? KSM : g.kscale.mode;
    const int kw = unit / g.nb, kblk = unit - kw * g.nb;

    auto load = [&](int kt, double (&qa)[4], double (&qb)[4], double (&qd)[NKS], int (&qc)[NKS]) {
        const int k0 = kbeg + kt * BK;
        tile_load<TA, FULL>(A, g.A.ld, m0, g.M, k0, g.K, tid, qa);
        tile_load<!TB, FULL>(B, g.B.ld, n0, g.N, k0, g.K, tid, qb);
        if (ksmode != 0) {
            const int kb = TA ? k0 + ((tid & 3) << 2) : k0 + (tid >> 4);
#pragma unroll
            for (int i = 0; i < NKS; ++i) {
                const int kk = min(kb + i, g.K - 1);  // clamped: rows past K hold zeros in the tile anyway
                // (the mode is a compile-time constant in the full-tile launches: with the run-time choice between the two
                // sources, and between the four array modes when the factor is applied, the requests of a k-tile sat in
                // branches and were waited for on the spot)
                if (ksmode == 2) qc[i] = g.kscale.conf[(long)kw * g.kscale.conf_stride + kk];
                else qd[i] = g.kscale.d[(long)unit * g.kscale.stride + kk];
            }
        }
    };
    auto ksfactor = [&](double d, int c) -> double {
        if (ksmode == 2) return c > 0 ? g.kscale.cpos[kblk] : g.kscale.cneg[kblk];
        if (ksmode == 1) return d;
        if (ksmode == 3) return 1.0 / d;
        if (ksmode == 4) return fmin(1.0, d);
        return 1.0 / fmax(1.0, d);
    };
    auto store = [&](int buf, double (&qa)[4], const double (&qb)[4], const double (&qd)[NKS], const int (&qc)[NKS]) {
        if (ksmode != 0) {
            if (!TA) {
                const double sc = ksfactor(qd[0], qc[0]);
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[i] *= sc;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[i] *= ksfactor(qd[i], qc[i]);
            }
        }
        tile_store<TA>(As[buf], tid, qa);
        tile_store<!TB>(Bs[buf], tid, qb);
    };
    auto compute = [&](int cur) {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int kq = kk + lq, rot = kk;  // rows kk .. kk + 3 share (k >> 2) = kk >> 2: rotation 4 (kk >> 2) = kk
            const double a0 = As[cur][kq][(wm + li + rot) & 63], a1 = As[cur][kq][(wm + 16 + li + rot) & 63];
            const double b0 = Bs[cur][kq][(wn + li + rot) & 63], b1 = Bs[cur][kq][(wn + 16 + li + rot) & 63];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
    };

    // The two workgroups that share a CU (grid = 2 per CU at 32 units) run the same program: left alone they reach
    // their MFMA bursts and their barriers together.  The second half of the grid starts half a k-step late.
    if (gemm_stagger && blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_sleep(16);
    load(0, ra[0], rb[0], ksd[0], ksc[0]);
    store(0, ra[0], rb[0], ksd[0], ksc[0]);
    GEMM_LDS_BARRIER();
    // (the requests are unconditional - past the last tile they fetch it again, harmlessly: behind `if (kt + 2 < nk)` the
    // compiler cannot count the requests in flight at the merge and waits for ALL of them before a tile goes to LDS,
    // i.e. also for the tile requested a moment ago - the ring was one tile deep, not two)
    load(min(1, nk - 1), ra[1], rb[1], ksd[1], ksc[1]);
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile kt: LDS buffer 0, its registers (slot 0) are free for tile kt+2
        load(min(kt + 2, nk - 1), ra[0], rb[0], ksd[0], ksc[0]);
        compute(0);
        if (kt + 1 < nk) store(1, ra[1], rb[1], ksd[1], ksc[1]);
        GEMM_LDS_BARRIER();
        if (kt + 1 >= nk) break;
        // odd tile kt+1: LDS buffer 1, slot 1 free for tile kt+3
        load(min(kt + 3, nk - 1), ra[1], rb[1], ksd[1], ksc[1]);
        compute(1);
        if (kt + 2 < nk) store(0, ra[0], rb[0], ksd[0], ksc[0]);
        GEMM_LDS_BARRIER();
    }

    // epilogue: lane holds C[m = .. + li][n = .. + lq + 4r]
    const bool plain = g.rowscale.mode == 0 && g.colscale.mode == 0 && g.adddiag.mode == 0 && m0 + BM <= g.M &&
                       n0 + BN <= g.N;
    if (plain) {  // the common case (chain products, UDT GEMMs): no lookups, no bounds tests
        double cold[2][2][4];
        if (g.beta) {  // accumulate: C requested first, all sixteen elements in one batch (was: load, wait, add, store per element)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cold[ti][tj][r] = C[(long)g.ldc * (n0 + wn + tj * 16 + lq + 4 * r) + m0 + wm + ti * 16 + li];
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int m = m0 + wm + ti * 16 + li;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn + tj * 16 + lq + 4 * r;
                    double v = acc[ti][tj][r] * g.alpha;
                    if (m == n) v += g.ident;
                    if (g.beta) v += cold[ti][tj][r];
                    C[(long)g.ldc * n + m] = v;
                }
            }
        }
        return;
    }
    // Scales that are plain arrays (mode 0 / 1: the products of calculate_greens_AVX!, stack.jl:346-348, :362-368): every
    // lookup is requested up front with a clamped index, one batch, no branch per value - behind the run-time mode dispatch of
    // vs_get the 8 + 2 + 2 loads of a lane went out one after the other, each waited for (tools/scan_isa.py: runs of 21)
    if (g.rowscale.mode <= 1 && g.colscale.mode <= 1 && g.adddiag.mode <= 1) {
        const double *csd = g.colscale.mode ? g.colscale.d + (long)unit * g.colscale.stride : nullptr;
        const double *rsd = g.rowscale.mode ? g.rowscale.d + (long)unit * g.rowscale.stride : nullptr;
        const double *add = g.adddiag.mode ? g.adddiag.d + (long)unit * g.adddiag.stride : nullptr;
        double csv[2][4] = {{1.0, 1.0, 1.0, 1.0}, {1.0, 1.0, 1.0, 1.0}}, rsv[2] = {1.0, 1.0}, adv[2] = {0.0, 0.0}, cold[2][2][4];
        const int mc0 = min(m0 + wm + li, g.M - 1), mc1 = min(m0 + wm + 16 + li, g.M - 1);
        if (csd) {  // (one uniform branch around each batch: a select per value made the compiler branch per value)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) csv[tj][r] = csd[min(n0 + wn + tj * 16 + lq + 4 * r, g.N - 1)];
        }
        if (rsd) { rsv[0] = rsd[mc0]; rsv[1] = rsd[mc1]; }
        if (add) { adv[0] = add[mc0]; adv[1] = add[mc1]; }
        if (g.beta) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = min(m0 + wm + ti * 16 + li, g.M - 1), n = min(n0 + wn + tj * 16 + lq + 4 * r, g.N - 1);
                        cold[ti][tj][r] = C[(long)g.ldc * n + m];
                    }
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int m = m0 + wm + ti * 16 + li;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn + tj * 16 + lq + 4 * r;
                    double v = acc[ti][tj][r];
                    if (g.row_first) { v *= rsv[ti]; v *= csv[tj][r]; } else { v *= csv[tj][r]; v *= rsv[ti]; }
                    v *= g.alpha;
                    if (m == n) v += g.ident + adv[ti];
                    if (g.beta) v += cold[ti][tj][r];
                    if (m < g.M && n < g.N) C[(long)g.ldc * n + m] = v;
                }
        }
        return;
    }
    double csv[2][4];  // column scales of my 8 columns, looked up once
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn + tj * 16 + lq + 4 * r;
            csv[tj][r] = (g.colscale.mode && n < g.N) ? vs_get(g.colscale, unit, g.nb, n) : 1.0;
        }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int m = m0 + wm + ti * 16 + li;
        if (m >= g.M) continue;
        const double rs = g.rowscale.mode ? vs_get(g.rowscale, unit, g.nb, m) : 1.0;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn + tj * 16 + lq + 4 * r;
                if (n >= g.N) continue;
                double v = acc[ti][tj][r];
                const double cs = csv[tj][r];
                if (g.row_first) { v *= rs; v *= cs; } else { v *= cs; v *= rs; }
                v *= g.alpha;
                if (m == n) {
                    v += g.ident;
                    if (g.adddiag.mode) v += vs_get(g.adddiag, unit, g.nb, m);
                }
                double *c = C + (long)g.ldc * n + m;
                if (g.beta) v += *c;
                *c = v;
            }
        }
    }
}

hipError_t launch_gemm(const GemmArgs &g, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const int groups = (g.n_units + 7) / 8;
    dim3 grid(groups * 8 * tm * tn), block(256);
    const int stagger = kernel_switches().gemm_stagger;
    const bool full = g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0;
#define GEMM_LAUNCH4(TA, TB, FU, KS) \
    hipExtLaunchKernelGGL((gemm_kernel<TA, TB, FU, KS>), grid, block, 0, s, start, stop, 0, g, tm, tn, stagger)
#define GEMM_LAUNCH(TA, TB)                                                        \
    do {                                                                           \
        if (!full) GEMM_LAUNCH4(TA, TB, false, -1);                                \
        else if (g.kscale.mode == 0) GEMM_LAUNCH4(TA, TB, true, 0);                \
        else if (g.kscale.mode == 2) GEMM_LAUNCH4(TA, TB, true, 2);                \
        else if (g.kscale.mode == 1) GEMM_LAUNCH4(TA, TB, true, 1);                \
        else if (g.kscale.mode == 3) GEMM_LAUNCH4(TA, TB, true, 3);                \
        else if (g.kscale.mode == 4) GEMM_LAUNCH4(TA, TB, true, 4);                \
        else if (g.kscale.mode == 5) GEMM_LAUNCH4(TA, TB, true, 5);                \
        else GEMM_LAUNCH4(TA, TB, true, -1);                                       \
    } while (0)
    if (g.transA) {
        if (g.transB) GEMM_LAUNCH(true, true);
        else GEMM_LAUNCH(true, false);
    } else {
        if (g.transB) GEMM_LAUNCH(false, true);
        else GEMM_LAUNCH(false, false);
    }
#undef GEMM_LAUNCH
#undef GEMM_LAUNCH4
    return hipGetLastError();
}

// Flush of the delayed Sherman-Morrison updates: C += U V' with K = 64 slots (sweep.hip keeps
// U'[t][m] at U[t + n m] and V[m][t] at VT[t + n m]).  The whole K extent of both operands fits LDS at
// once (2 x 40 KB), so there is one barrier instead of the generic kernel's four pipelined k-tiles, and
// the C tile is requested before anything else: the kernel is bound by that read-modify-write.
// Requires n % 64 == 0; same tile -> wave -> lane mapping and XCD-aware block map as gemm_kernel.
constexpr int FK = 64;
__global__ __launch_bounds__(256) void gemm_flush_kernel(int n, int n_units, const double *__restrict__ Uall,
                                                        const double *__restrict__ VTall, long sUV,
                                                        double *__restrict__ Call, long sC, int tiles_m, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    double(*As)[LS] = reinterpret_cast<double(*)[LS]>(fsm);             // [FK][LS]: U tile, m contiguous
    double(*Bs)[LS] = reinterpret_cast<double(*)[LS]>(fsm + FK * LS);   // [FK][LS]: V' tile, n contiguous
    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int T = tiles_m * tiles_n;
    const int unit = (seq / T) * 8 + xcd;
    if (unit >= n_units) return;
    const int tile = seq % T;
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const double *__restrict__ U = Uall + (long)unit * sUV;
    const double *__restrict__ VT = VTall + (long)unit * sUV;
    double *__restrict__ C = Call + (long)unit * sC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int li = lane & 15, lq = lane >> 4;

    // C tile first: lane holds C[m = .. + li][n = .. + lq + 4r]
    d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ti][tj][r] = C[(long)n * (n0 + wn + tj * 16 + lq + 4 * r) + m0 + wm + ti * 16 + li];
    // operands: 4 passes of 16 k-rows; thread -> k = pass*16 + tid/16, 4 consecutive m (resp. n)
    const int kr = tid >> 4, c4 = (tid & 15) << 2;
    double ra[4][4], rb[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double *qa = U + (long)n * (p * 16 + kr) + m0 + c4;
        const double *qb = VT + (long)n * (p * 16 + kr) + n0 + c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[p][i] = qa[i]; rb[p][i] = qb[i]; }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double *da = &As[p * 16 + kr][c4], *db = &Bs[p * 16 + kr][c4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { da[i] = ra[p][i]; db[i] = rb[p][i]; }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FK; kk += 4) {
        const int kq = kk + lq;
        const double a0 = As[kq][wm + li], a1 = As[kq][wm + 16 + li];
        const double b0 = Bs[kq][wn + li], b1 = Bs[kq][wn + 16 + li];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)n * (n0 + wn + tj * 16 + lq + 4 * r) + m0 + wm + ti * 16 + li] = acc[ti][tj][r];
}
hipError_t launch_gemm_flush(int n, int n_units, const double *U, const double *VT, long sUV, double *C, long sC,
                             hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    if (n % 64 != 0) return hipErrorInvalidValue;
    const int tm = n / BM, tn = n / BN;
    const int groups = (n_units + 7) / 8;
    const size_t lds = 2 * (size_t)FK * LS * sizeof(double);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device (function attributes are per device)
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)gemm_flush_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_mask |= 1u << dev;
    }
    hipExtLaunchKernelGGL(gemm_flush_kernel, dim3(groups * 8 * tm * tn), dim3(256), lds, s, start, stop, 0, n, n_units, U,
                          VT, sUV, C, sC, tm, tn);
    return hipGetLastError();
}

// fp64 MFMA peak probe: independent accumulators, no memory traffic.
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, double *sink)
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
hipError_t launch_mfma_peak(int iters, int blocks, double *sink, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, iters, sink);
    return hipGetLastError();
}

}  // namespace dqmc
