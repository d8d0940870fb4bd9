// sweep_lu.hip — sweep_spatial (src/flavors/DQMC/DQMC.jl:546-582) with propose_local /
// accept_local! of HubbardModelAttractive.jl:113-155 and HubbardModelRepulsive.jl:128-232 as a
// "decide on the 64 x 64 block, apply with MFMA" pair of kernels per chunk of 64 sites.
//
// Within a chunk c of 64 consecutive sites the Metropolis decisions only ever look at the diagonal of
// S = G[c, c], and an accepted flip at site s changes the later part of that block by the rank-1 term
//     S[k, l] += x_s S[k, s] S[s, l]      (k, l > s;  x_s = gamma / (1 + gamma (1 - S[s, s])))
// (accept_local!'s  G -= (e_i - G[:, i]) x G[i, :]  restricted to the chunk).  The sequential part is
// therefore a conditional LU-type elimination of a 64 x 64 matrix:
//
//   sweep_lu_kernel      one wave per walker.  S and S' live in MFMA accumulator layout (16 x 16 tiles of
//                        v_mfma_f64_16x16x4_f64), so that row s of S / of S' is at the same time the B operand
//                        (the row G[s, :]) and the A operand (the column G[:, s]) of the rank-1 update: no data
//                        moves between lanes, the diagonal entry is read with v_readlane.  Tiles of the current
//                        block row are updated at every accepted site, the remaining ones once per panel of four
//                        sites with a full k = 4 MFMA.  By-products: the unit-triangular inverses of the 16 x 16
//                        diagonal blocks (one more MFMA per site each).
//   sweep_flush_lu_kernel  G_out = G_in + T R0,  R0 = G_in[c, :],  T = (G_in[:, c] - E) X (I - Uu X)^-1 (I - L X)^-1
//                        with Uu / L the strict upper / lower parts of the eliminated block and X = diag(x): two
//                        block-triangular solves per 16 columns of T' (MFMA, operands of the triangles straight
//                        from the first kernel's register images) and the K = 64 update of a 64 x 64 tile of G.
//
// This is a re-association of the reference's arithmetic only (the literal algorithm is restated in
// tools/proto/lu_sweep_proto.py and in the oracle).
#include "kernels.h"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace dqmc {

typedef double d4 __attribute__((ext_vector_type(4)));

// per-unit image written by sweep_lu_kernel, read by sweep_flush_lu_kernel (doubles):
//   [U pair tiles 6][L pair tiles 6][PT 4][Q 4] each 4 regs x 64 lanes, then x[64]
constexpr int LU_TILE = 256;
constexpr int LU_OFF_U = 0, LU_OFF_L = 6 * LU_TILE, LU_OFF_PT = 12 * LU_TILE, LU_OFF_Q = 16 * LU_TILE;
constexpr int LU_IMG = 20 * LU_TILE;
constexpr int LU_STRIDE = LU_IMG + 64;
__host__ __device__ constexpr int lu_pair(int K, int J)  // K < J: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
{
    return K == 0 ? J - 1 : (K == 1 ? J + 1 : 5);
}
size_t sweep_lu_image_doubles() { return LU_STRIDE; }

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// Pointers into device memory, typed as such.  Inside a __noinline__ device function, or when a pointer reaches the kernel
// inside a by-value struct, the compiler only knows a generic pointer: every access becomes a FLAT operation, which counts
// in vmcnt AND lgkmcnt (each wait for an LDS read then also waits for the global requests in flight) and may alias LDS
// (a load followed by an LDS store is then waited for on the spot: the staging loops of the flush workgroups inside the
// fused launch were 27 round trips one after the other).  With the address space in the type they are global_load /
// global_store again.
#define DQMC_GLOBAL __attribute__((address_space(1)))
typedef const DQMC_GLOBAL double *gcdp;
typedef DQMC_GLOBAL double *gdp;
typedef double d2v __attribute__((ext_vector_type(2)));  // (HIP's double2 is a class: no assignment from address space 1)
typedef const DQMC_GLOBAL d2v *gcd2p;

// Accumulator layout of v_mfma_f64_16x16x4_f64: register r of lane l = (g = l >> 4, ci = l & 15) holds element
// [4 r + g][ci] of a 16 x 16 tile; the A operand is A[i = ci][k = g], the B operand B[k = g][j = ci].
// S[I][J] (J >= I): tile of S, rows 16 I.., columns 16 J..;  ST[I][J]: tile of S', i.e. ST[I][J][rho][kappa] =
// S[16 J + kappa][16 I + rho].  Only the upper block triangle of both is kept.

// ---------------------------------------------------------------------------------------------------------
// The same elimination on the four SIMDs of a CU: wave J owns block column J of S and of S' (tiles (I, J), I <= J).
// In block I0 wave I0 is the PIVOT: it runs the decisions on its diagonal tiles and publishes, per accepted site,
// the two A operands (x G[:, s], x G[s, :] on the block's rows) through LDS; waves J > I0 apply them to their strip
// tiles (I0, J), exchange the finished strip rows once per panel and do the k = 4 updates of their later tiles;
// wave (I0 + 2) % 4 also carries the two triangular inverses of the diagonal block.  Each slot of the step ring is
// written once per launch (no reuse), flags are LDS words with workgroup-scope release / acquire; every wait is
// bounded (a time-out sets *errflag and lets all waves run through).
constexpr int LU4_SPIN = 1 << 21;
template <int NB>
struct Lu4Smem {
    double2 aST[NB][64][64];     // [block][site][lane]: the two A operands of an accepted site, x * (column s on the
                                 // block rows) for the S update and x * (row s on the block columns) for the S' update
    double xs[NB][64];           // x of every site (0 = rejected), zeroed at kernel start
    double strip[12][NB][2][64]; // finished strip registers of (I0, panel, wave): S row panel, S' row panel
    double negv[64];
    int step[64];                // 0: not decided; else (ndraw << 3) | (exhausted << 2) | (2: accepted, 1: rejected)
    int sflag[12];
    unsigned acc16[4], neg16[4];
    int ndraw[4], exh[4];
    int abort;
};
__host__ __device__ constexpr int lu4_strip_idx(int I0, int r0, int I) { return I0 == 0 ? r0 * 2 + (I - 1) : 8 + r0; }

// LDS executes the DS instructions of one wave in issue order, so a flag written (read) after its payload in program
// order needs no s_waitcnt in between; the empty asm statements only pin the program order for the compiler.
#define LU4_ORDER() asm volatile("" ::: "memory")

// wave-uniform wait on an LDS word: the loaded value is made scalar right away, so that the loop and everything that
// depends on the flag stays on the scalar unit
__device__ __forceinline__ int lu4_wait(int *flag, int *abortf)
{
    int v = 0;
    for (int it = 0; it < LU4_SPIN; ++it) {
        v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v) break;
        if ((it & 63) == 63 &&
            __builtin_amdgcn_readfirstlane(__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))
            return -1;
        __builtin_amdgcn_s_sleep(1);
    }
    if (!v) {
        __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return -1;
    }
    LU4_ORDER();
    return v;
}
// 1 / r to working precision: v_rcp_f64 and two Newton steps (the correctly rounded quotient is not needed: x only
// enters the update of G, never a decision)
__device__ __forceinline__ double lu4_rcp(double r)
{
    double y = __builtin_amdgcn_rcp(r);
    double e = __builtin_fma(-r, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-r, y, 1.0);
    return __builtin_fma(y, e, y);
}

extern __shared__ __attribute__((aligned(16))) double lu4_lds[];
#ifdef LU4_STAMPS  // diagnostic build only (tools/lu4_stamps.py): cycle stamps of walker 0 into a buffer of their own
__device__ long long *lu4_stamp_ptr = nullptr;
// LU4_STAMPS == 2: only the coarse stamps (index >= 320: phases of a wave), which do not disturb the site loop
#define LU4_STAMP(idx)                                                                       \
    do {                                                                                     \
        if ((LU4_STAMPS != 2 || (idx) >= 320) && w == 0 && (threadIdx.x & 63) == 0 && lu4_stamp_ptr)            \
            lu4_stamp_ptr[(idx)] = (long long)__builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define LU4_STAMP(idx) do { } while (0)
#endif
#ifdef LU4_STAMPS
#ifndef FL_STAMP_BID
#define FL_STAMP_BID 1200
#endif
#define FL_STAMP(k)                                                                                    \
    do {                                                                                               \
        if (bid == FL_STAMP_BID && threadIdx.x == 0 && lu4_stamp_ptr)                                 \
            lu4_stamp_ptr[480 + (k)] = (long long)__builtin_amdgcn_s_memtime();                       \
    } while (0)
#else
#define FL_STAMP(k) do { } while (0)
#endif
// one function per wave role (not inlined into each other: each gets its own register allocation)
// PRO: the previous chunk (sites site0p ..+63, images imgp_all) has not been applied to G yet: the wave adds its
// contribution T R0 to its tiles itself (the same two block-triangular solves and K = 64 products as
// sweep_flush_lu_kernel, for the 16 rows / columns of its block column), so that this elimination can run beside the
// flush of the previous chunk instead of behind it.
template <int NB, bool FULL, bool PRO, int J>
__device__ __noinline__ void lu4_wave(int n, const double *__restrict__ Gall, long strideG, int w, int site0,
                                      int nsites, double *__restrict__ img_all, const SweepConsts &sc,
                                      unsigned long long cbits_v, double uvec, int check_sign, int site0p,
                                      const double *__restrict__ imgp_all)
{
    Lu4Smem<NB> &sm = *reinterpret_cast<Lu4Smem<NB> *>(lu4_lds);
    const int lane = threadIdx.x & 63, g = lane >> 4, ci = lane & 15;
    const unsigned long long cbits =
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cbits_v >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(cbits_v & 0xffffffffull));
    d4 S[NB][J + 1], ST[NB][J + 1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const gcdp G = (gcdp)(Gall + (long)(w * NB + b) * strideG + (long)site0 * n + site0);
#pragma unroll
        for (int I = 0; I <= J; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rho = 16 * I + 4 * r + g, kap = 16 * J + ci;  // S[rho][kap]; ST: S[kap][rho]
                if (FULL) {
                    S[b][I][r] = G[rho + (long)n * kap];
                    ST[b][I][r] = G[kap + (long)n * rho];
                } else {
                    const bool ok = rho < nsites && kap < nsites;
                    const int rc = min(rho, nsites - 1), kc = min(kap, nsites - 1);
                    const double a = G[rc + (long)n * kc], c = G[kc + (long)n * rc];
                    S[b][I][r] = ok ? a : 0.0;
                    ST[b][I][r] = ok ? c : 0.0;
                }
            }
    }
    if (PRO) {
        LU4_STAMP(400 + 8 * J + 0);
        // EARLY (one block per walker, images staged in LDS): wave 0 - the first pivot - only adds what its own diagonal
        // tile needs (two tile products) and starts the site loop, while waves 1..3 finish the block update behind
        // its first sixteen decisions; the products of block row 0 that wave 0 used to make for them (S tiles (0, Jc))
        // are made by their owners from wave 0's T' tiles, which it leaves in LDS.  The scratch must then stay clear of
        // the ring slots the first pivots write: the R0 block [64 t'][66] sits on slots 31..63 (exactly 33 slots;
        // a pivot of block >= 1 waits for the three "products done" flags before it writes its first slot), deltas,
        // wave 0's T' and the flags behind the staged images.  Otherwise (NB == 2: no LDS left) everything is aliased
        // onto the not yet used ring and all four waves meet at two barriers before the first decision.
        constexpr bool EARLY = (NB == 1) && (sizeof(Lu4Smem<NB>) + LU_STRIDE * sizeof(double) <= 150 * 1024);
        double *xtra = lu4_lds + (sizeof(Lu4Smem<NB>) + 15) / 16 * 2 + LU_STRIDE;  // [deltas 1536][T' of wave 0 1024][flags]
        double *Rl = EARLY ? reinterpret_cast<double *>(&sm.aST[0][31][0]) : reinterpret_cast<double *>(&sm.aST[0][0][0]);
        double *ttx = EARLY ? xtra : Rl + 64 * 66;
        double *tt0 = xtra + 1536;
        int *pflag = reinterpret_cast<int *>(xtra + 1536 + 1024);  // [0] T' of wave 0 published, [J] deltas of wave J
                                                                   // written, [4 + J] wave J is done with the scratch
        const int tid = threadIdx.x;
#pragma unroll 1
        for (int b = 0; b < NB; ++b) {
            const gcdp Gin = (gcdp)(Gall + (long)(w * NB + b) * strideG);
            const gcdp imgp = (gcdp)(imgp_all + (long)(w * NB + b) * LU_STRIDE);
            const int t = site0 + 16 * J + ci;
            d4 az[4];
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) az[Jb][r] = Gin[t + (long)n * (site0p + 16 * Jb + 4 * r + g)];
            // R0 block: thread -> column t' = tid / 4 of this chunk, 16 consecutive sites of the previous one
            d2v r0v[8];
            {
                const int tl = tid >> 2, s0 = (tid & 3) * 16;
                const gcd2p q = reinterpret_cast<gcd2p>(Gin + (long)n * (site0 + tl) + site0p + s0);
#pragma unroll
                for (int i = 0; i < 8; ++i) r0v[i] = q[i];
            }
            // the images of the previous chunk (the 80 operands of the two triangles, x): NB == 1 stages them in LDS
            // behind the shared structure, all four waves copying together, so that the whole prologue needs ONE memory
            // round trip (G columns, R0 block and images in flight together) and no operand lives in a register
            // before its MFMA; NB == 2 has no LDS left for that and requests the operands into registers
            constexpr bool STAGE = sizeof(Lu4Smem<NB>) + LU_STRIDE * sizeof(double) <= 150 * 1024;
            double *imgl = lu4_lds + (sizeof(Lu4Smem<NB>) + 15) / 16 * 2;
            double opU[STAGE ? 1 : 6][4], opL[STAGE ? 1 : 6][4], opP[STAGE ? 1 : 4][4], opQ[STAGE ? 1 : 4][4], xr[4][4];
            if (STAGE) {
                constexpr int NI = (LU_STRIDE / 2 + 255) / 256;
                const gcd2p src = reinterpret_cast<gcd2p>(imgp);
                d2v *dst = reinterpret_cast<d2v *>(imgl);
                d2v iv[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) iv[i] = src[min(tid + 256 * i, LU_STRIDE / 2 - 1)];
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (tid + 256 * i < LU_STRIDE / 2) dst[tid + 256 * i] = iv[i];
            } else {
#pragma unroll
                for (int pr = 0; pr < (STAGE ? 0 : 6); ++pr)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        opU[pr][q] = imgp[LU_OFF_U + pr * LU_TILE + q * 64 + lane];
                        opL[pr][q] = imgp[LU_OFF_L + pr * LU_TILE + q * 64 + lane];
                    }
#pragma unroll
                for (int Jb = 0; Jb < (STAGE ? 0 : 4); ++Jb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        opP[Jb][q] = imgp[LU_OFF_PT + Jb * LU_TILE + q * 64 + lane];
                        opQ[Jb][q] = imgp[LU_OFF_Q + Jb * LU_TILE + q * 64 + lane];
                    }
            }
            {   // (all loads of the prologue are in flight by now: one round trip)
                const int tl = tid >> 2, s0 = (tid & 3) * 16;
                d2v *d = reinterpret_cast<d2v *>(Rl + tl * 66 + s0);
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = r0v[i];
            }
            if (STAGE) __syncthreads();  // images (and the R0 block) are in LDS
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
                for (int q = 0; q < 4; ++q) xr[Jb][q] = STAGE ? imgl[LU_IMG + 16 * Jb + 4 * q + g] : imgp[LU_IMG + 16 * Jb + 4 * q + g];
#define PRO_U(pr, q) (STAGE ? imgl[LU_OFF_U + (pr) * LU_TILE + (q) * 64 + lane] : opU[STAGE ? 0 : (pr)][q])
#define PRO_L(pr, q) (STAGE ? imgl[LU_OFF_L + (pr) * LU_TILE + (q) * 64 + lane] : opL[STAGE ? 0 : (pr)][q])
#define PRO_P(jb, q) (STAGE ? imgl[LU_OFF_PT + (jb) * LU_TILE + (q) * 64 + lane] : opP[STAGE ? 0 : (jb)][q])
#define PRO_Q(jb, q) (STAGE ? imgl[LU_OFF_Q + (jb) * LU_TILE + (q) * 64 + lane] : opQ[STAGE ? 0 : (jb)][q])
            d4 xz[4], tt[4];
#if defined(LU4_STAMPS) && LU4_STAMPS != 2
            LU4_STAMP(400 + 8 * J + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LU4_STAMP(400 + 8 * J + 2);
#endif
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb) {
#pragma unroll
                for (int K = 0; K < Jb; ++K)
#pragma unroll
                    for (int q = 0; q < 4; ++q) az[Jb] = MFMA(PRO_U(lu_pair(K, Jb), q), xz[K][q], az[Jb]);
                d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) z = MFMA(PRO_P(Jb, q), az[Jb][q], z);
#pragma unroll
                for (int r = 0; r < 4; ++r) xz[Jb][r] = z[r] * xr[Jb][r];
            }
#pragma unroll
            for (int Jb = 3; Jb >= 0; --Jb) {
                d4 a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int K = Jb + 1; K < 4; ++K)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a = MFMA(PRO_L(lu_pair(Jb, K), q), tt[K][q], a);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = xz[Jb][r] + xr[Jb][r] * a[r];
                d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) o = MFMA(PRO_Q(Jb, q), a[q], o);
                tt[Jb] = o;
            }
#undef PRO_U
#undef PRO_L
#undef PRO_P
#undef PRO_Q
            LU4_STAMP(400 + 8 * J + 3);
            if (EARLY) {
                double *dlt = ttx;  // [pair (J, Jc)][4 regs][64 lanes]
                if (J == 0) {
#pragma unroll
                    for (int K = 0; K < 4; ++K)
#pragma unroll
                        for (int q = 0; q < 4; ++q) tt0[(K * 4 + q) * 64 + lane] = tt[K][q];
                    LU4_ORDER();
                    __hip_atomic_store(&pflag[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (int K = 0; K < 4; ++K)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int sk = 16 * K + 4 * q + g;
                            ST[b][0] = MFMA(Rl[ci * 66 + sk], tt[K][q], ST[b][0]);
                            S[b][0] = MFMA(tt[K][q], Rl[ci * 66 + sk], S[b][0]);
                        }
                } else {
                    // own T' rows: S' tiles (I, J), I <= J; S tile (J, J); S tiles (J, Jc), Jc > J, as deltas for wave Jc
#pragma unroll
                    for (int K = 0; K < 4; ++K)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int sk = 16 * K + 4 * q + g;
#pragma unroll
                            for (int I = 0; I <= J; ++I) ST[b][I] = MFMA(Rl[(16 * I + ci) * 66 + sk], tt[K][q], ST[b][I]);
                            S[b][J] = MFMA(tt[K][q], Rl[(16 * J + ci) * 66 + sk], S[b][J]);
                        }
#pragma unroll
                    for (int Jc = J + 1; Jc < 4; ++Jc) {
                        d4 dl = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int K = 0; K < 4; ++K)
#pragma unroll
                            for (int q = 0; q < 4; ++q) dl = MFMA(tt[K][q], Rl[(16 * Jc + ci) * 66 + 16 * K + 4 * q + g], dl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) dlt[(lu_pair(J, Jc) * 4 + r) * 64 + lane] = dl[r];
                    }
                    LU4_ORDER();
                    __hip_atomic_store(&pflag[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // S tile (0, J) of block row 0 from wave 0's T' tiles
                    if (lu4_wait(&pflag[0], &sm.abort) > 0) {
#pragma unroll
                        for (int K = 0; K < 4; ++K)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                S[b][0] = MFMA(tt0[(K * 4 + q) * 64 + lane], Rl[(16 * J + ci) * 66 + 16 * K + 4 * q + g], S[b][0]);
                    }
#pragma unroll
                    for (int I = 1; I < J; ++I)
                        if (lu4_wait(&pflag[I], &sm.abort) > 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) S[b][I][r] += dlt[(lu_pair(I, J) * 4 + r) * 64 + lane];
                        }
                    LU4_ORDER();
                    __hip_atomic_store(&pflag[4 + J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                continue;
            }
            // Block update, five tile products of 16 MFMAs on every wave: wave J computes what needs ITS OWN T' rows -
            // the S' tiles (I, J), I <= J, of its block column, and the S tiles (J, Jc), Jc >= J, of block ROW J.  The
            // latter belong to wave Jc for Jc > J: they go there as deltas through LDS.  (Ownership by block column
            // alone gave wave 3 eight products and wave 0 two, and needed every wave's T' in LDS.)
            if (!STAGE) __syncthreads();  // the R0 block is in LDS (STAGE: covered by the barrier behind the staging)
            LU4_STAMP(400 + 8 * J + 4);
            double *dlt = ttx;  // [pair (J, Jc)][4 regs][64 lanes]
#pragma unroll
            for (int K = 0; K < 4; ++K)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sk = 16 * K + 4 * q + g;
#pragma unroll
                    for (int I = 0; I <= J; ++I)  // S' tile (I, J): rows of G in my block column, columns in block I
                        ST[b][I] = MFMA(Rl[(16 * I + ci) * 66 + sk], tt[K][q], ST[b][I]);
                    S[b][J] = MFMA(tt[K][q], Rl[(16 * J + ci) * 66 + sk], S[b][J]);  // S tile (J, J)
                }
#pragma unroll
            for (int Jc = J + 1; Jc < 4; ++Jc) {  // S tile (J, Jc): rows in my block, columns in block Jc
                d4 dl = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int K = 0; K < 4; ++K)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dl = MFMA(tt[K][q], Rl[(16 * Jc + ci) * 66 + 16 * K + 4 * q + g], dl);
#pragma unroll
                for (int r = 0; r < 4; ++r) dlt[(lu_pair(J, Jc) * 4 + r) * 64 + lane] = dl[r];
            }
            __syncthreads();  // the deltas of the waves to my left are in LDS
#pragma unroll
            for (int I = 0; I < J; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[b][I][r] += dlt[(lu_pair(I, J) * 4 + r) * 64 + lane];
            __syncthreads();  // scratch is free again (next block / the step ring)
        }
        LU4_STAMP(320 + 8 * J + 7);
    }
    LU4_STAMP(320 + 8 * J + 0);
#if defined(LU4_STAMPS) && LU4_STAMPS != 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LU4_STAMP(320 + 8 * J + 1);
#endif
    // model constants as scalars (arguments of a non-inlined function arrive in vector registers): the per-site
    // selects then run on the scalar unit
    auto sgpr = [](double v) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffull));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double g0 = sgpr(sc.gamma[0]), g1 = sgpr(sc.gamma[1]), e0 = sgpr(sc.ebos[0]), e1 = sgpr(sc.ebos[1]);
    const double du0 = sgpr(sc.dup[0]), du1 = sgpr(sc.dup[1]), dd0 = sgpr(sc.ddn[0]), dd1 = sgpr(sc.ddn[1]);
    int lastflag = 0;  // the newest step word seen (carries the draw counter from pivot to pivot)

#pragma unroll
    for (int I0 = 0; I0 < 4; ++I0) {
        if (!FULL && 16 * I0 >= nsites) break;
        // the triangular inverses ride on a wave that is idle in this block (wave 0 after its own block, wave 1 in the
        // last one); in block 0 everybody is busy and they go to wave 3, whose lag has three blocks to drain - never to
        // the next pivot (I0 + 1), whose lag at the end of the block is the hand-over time (stamps: with the inverses of
        // block 0 on wave 2 that wave entered block 2 about 4000 cycles behind)
        const bool PIV = (I0 == J), HLP = (J > I0), PTQ = (J == (I0 == 0 ? 3 : (I0 == 3 ? 1 : 0)));
        if (!PIV && !HLP && !PTQ) continue;
        d4 PT[NB], Q[NB];
        if (PTQ) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) PT[b][r] = Q[b][r] = (4 * r + g == ci) ? 1.0 : 0.0;
        }
        int ndraw = (lastflag >> 3) & 0xfff, exhausted = (lastflag >> 2) & 1;
        unsigned accb = 0, negb = 0;
        double negv = 0.0;
        // The diagonal entry the next decision needs is carried as a scalar: S[s+1][s+1] + x S[s+1][s] S[s][s+1]
        // from registers that the MFMAs of site s have not touched yet, so a decision never waits for an MFMA.
        // helper side: flag and payload of the NEXT site are requested before the MFMAs of the current one, so that a
        // helper that is behind (after the prologue, after a panel end) works through the ring at MFMA pace instead of
        // one LDS round trip per site; a request that came too early (flag not yet set) is repeated by the spin loop
        int pf_v = 0;
        double2 pf_pay[NB];
        double pf_xh[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { pf_pay[b] = make_double2(0.0, 0.0); pf_xh[b] = 0.0; }
        double dcur[NB];
        if (PIV) {
            if (PRO && NB == 1 && I0 >= 1 && sizeof(Lu4Smem<NB>) + LU_STRIDE * sizeof(double) <= 150 * 1024) {
                // EARLY prologue: the R0 block lies on ring slots 31..63 until waves 1..3 are done with it
                int *pflag = reinterpret_cast<int *>(lu4_lds + (sizeof(Lu4Smem<NB>) + 15) / 16 * 2 + LU_STRIDE + 1536 + 1024);
                (void)lu4_wait(&pflag[5], &sm.abort);
                (void)lu4_wait(&pflag[6], &sm.abort);
                (void)lu4_wait(&pflag[7], &sm.abort);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) dcur[b] = readlane_d(S[b][J][0], 0);
        }
#pragma unroll
        for (int r0 = 0; r0 < 4; ++r0) {
            double xv4[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) xv4[b] = 0.0;
            unsigned panel_acc = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 4 * r0 + ks, s = 16 * I0 + c;
                if (!FULL && s >= nsites) continue;
                if (PIV) {
                    LU4_STAMP(s);
                    const int spin = (int)((cbits >> s) & 1ull);
                    double det, p, xb[NB];
                    if (NB == 1) {  // HubbardModelAttractive.jl:113-127
                        const double gamma = spin ? g1 : g0;
                        const double r = 1.0 + gamma * (1.0 - dcur[0]);
                        det = r * r;
                        p = (spin ? e1 : e0) * det;
                        xb[0] = gamma * lu4_rcp(r);  // Attractive.jl:149: x = gamma / (1 + gamma * IG[i])
                    } else {        // HubbardModelRepulsive.jl:128-156,174-191
                        const double D0 = spin ? du1 : du0, D1 = spin ? dd1 : dd0;
                        const double R0 = 1.0 + D0 * (1.0 - dcur[0]), R1 = 1.0 + D1 * (1.0 - dcur[NB - 1]);
                        det = R0 * R1;
                        p = det;
                        const double inv_div = lu4_rcp(det);
                        xb[0] = (R1 * inv_div) * D0;
                        xb[NB - 1] = (R0 * inv_div) * D1;
                        if (check_sign && det < 0.0) {
                            negb |= 1u << c;
                            negv = lane == s ? det : negv;
                        }
                    }
                    bool acc;
                    if (p > 1.0) acc = true;  // DQMC.jl:573: rand() is consumed only when p <= 1
                    else {
                        const double u = readlane_d(uvec, ndraw);
                        ++ndraw;
                        if (u == 2.0) exhausted = 1;
                        acc = u < p;
                    }
                    const int word = (ndraw << 3) | (exhausted << 2);
                    const bool more = c < 15 && (FULL || s + 1 < nsites);
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int c1 = c + 1, l1 = 16 * (c1 & 3) + c1;  // lane of S[s+1][s+1] in register c1 >> 2
                    {
                        // branch-free in the decision: a rejected site runs the same stream with x = 0 (its two MFMAs add
                        // zero; a branch costs more in register copies at the join than the MFMA issue slots it saves)
                        const bool accu = __builtin_amdgcn_readfirstlane((int)acc) != 0;
                        accb |= (accu ? 1u : 0u) << c;
                        const bool lm2 = g == ks && ci > c;
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const double x = accu ? xb[b] : 0.0;
                            const double mS = lm2 ? S[b][J][r0] : 0.0, mT = lm2 ? ST[b][J][r0] : 0.0;
                            const double aS = x * mT, aT = x * mS;
                            if (more) {
                                // next diagonal entry S[s+1][s+1] + x S[s+1][s] S[s][s+1]: the product of the two
                                // off-diagonal entries and the old diagonal entry do not depend on this site's
                                // decision (only on the tile as site s - 1 left it), so they are taken out of the tile
                                // BEFORE x is known and the chain from decision to decision ends with one FMA
                                const double qs = readlane_d(mT * mS, 16 * ks + c1);
                                dcur[b] = __builtin_fma(x, qs, readlane_d(S[b][J][c1 >> 2], l1));
                            }
                            S[b][J] = MFMA(aS, mS, S[b][J]);
                            sm.aST[b][s][lane] = make_double2(aS, aT);
                            sm.xs[b][s] = x;
                            ST[b][J] = MFMA(aT, mT, ST[b][J]);
                        }
                        LU4_ORDER();
                        __hip_atomic_store(&sm.step[s], word | (accu ? 2 : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    lastflag = word;
                } else {
                    // flag and payload are requested together (in this order); the payload is only valid if the
                    // flag was already set
                    int v = c > 0 ? __builtin_amdgcn_readfirstlane(pf_v) : 0;
                    double2 pay[NB];
                    double xh[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) { pay[b] = pf_pay[b]; xh[b] = pf_xh[b]; }
                    for (int it = 0; v == 0 && it < LU4_SPIN; ++it) {
                        const int vr = __hip_atomic_load(&sm.step[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        LU4_ORDER();
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            pay[b] = sm.aST[b][s][lane];
                            xh[b] = sm.xs[b][s];
                        }
                        v = __builtin_amdgcn_readfirstlane(vr);
                        if (v) break;
                        if ((it & 63) == 63 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(
                                                   &sm.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
                            v = -1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (v == 0) {
                        __hip_atomic_store(&sm.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        v = -1;
                    }
                    if (v > 0) lastflag = v;
                    if (c < 15 && (FULL || s + 1 < nsites)) {  // next site of this block: request it now
                        pf_v = __hip_atomic_load(&sm.step[s + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        LU4_ORDER();
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            pf_pay[b] = sm.aST[b][s + 1][lane];
                            pf_xh[b] = sm.xs[b][s + 1];
                        }
                    }
                    LU4_STAMP(64 * (1 + J) + s);
                    if (v > 0 && (v & 3) == 2) {
                        panel_acc |= 1u << ks;
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const double aS = pay[b].x, aT = pay[b].y;
                            if (HLP) {
                                xv4[b] = g == ks ? xh[b] : xv4[b];
                                S[b][I0] = MFMA(aS, S[b][I0][r0], S[b][I0]);
                                ST[b][I0] = MFMA(aT, ST[b][I0][r0], ST[b][I0]);
                            }
                            if (PTQ) {
                                PT[b] = MFMA(aT, PT[b][r0], PT[b]);  // PT[j][:] += x G[s, j] PT[s][:]
                                Q[b] = MFMA(aS, Q[b][r0], Q[b]);     // Q[k][:]  += x G[k, s] Q[s][:]
                            }
                        }
                    }
                }
            }
            // panel end: the four sites applied to the later tiles of my block column with k = 4 MFMAs
            if (HLP && panel_acc != 0) {
                if (J < 3) {  // my finished strip rows for the waves to my right
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        sm.strip[lu4_strip_idx(I0, r0, J)][b][0][lane] = S[b][I0][r0];
                        sm.strip[lu4_strip_idx(I0, r0, J)][b][1][lane] = ST[b][I0][r0];
                    }
                    LU4_ORDER();
                    __hip_atomic_store(&sm.sflag[lu4_strip_idx(I0, r0, J)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {  // tile (J, J) first: it is the next pivot's diagonal tile
                    S[b][J] = MFMA(xv4[b] * ST[b][I0][r0], S[b][I0][r0], S[b][J]);
                    ST[b][J] = MFMA(xv4[b] * S[b][I0][r0], ST[b][I0][r0], ST[b][J]);
                }
#pragma unroll
                for (int I = I0 + 1; I < J; ++I) {
                    const int ok = lu4_wait(&sm.sflag[lu4_strip_idx(I0, r0, I)], &sm.abort);
                    if (ok > 0) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const double rs = sm.strip[lu4_strip_idx(I0, r0, I)][b][0][lane];
                            const double rt = sm.strip[lu4_strip_idx(I0, r0, I)][b][1][lane];
                            S[b][I] = MFMA(xv4[b] * rt, S[b][I0][r0], S[b][I]);
                            ST[b][I] = MFMA(xv4[b] * rs, ST[b][I0][r0], ST[b][I]);
                        }
                    }
                }
            }
        }
        // block row I0 is final: register images for the flush kernel
        LU4_STAMP(320 + 8 * J + 2 + I0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const gdp img = (gdp)(img_all + (long)(w * NB + b) * LU_STRIDE);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tpos = (ci >> 2) * 64 + 16 * (ci & 3) + 4 * r + g;
                if (HLP) {
                    img[LU_OFF_U + lu_pair(I0, J) * LU_TILE + r * 64 + lane] = S[b][I0][r];
                    img[LU_OFF_L + lu_pair(I0, J) * LU_TILE + tpos] = ST[b][I0][r];
                }
                if (PTQ) {
                    img[LU_OFF_PT + I0 * LU_TILE + tpos] = PT[b][r];
                    img[LU_OFF_Q + I0 * LU_TILE + r * 64 + lane] = Q[b][r];
                }
            }
            if (PIV && (lane >> 4) == I0) img[LU_IMG + lane] = sm.xs[b][lane];
        }
        if (PIV) {
            if (lane == 0) {
                sm.acc16[I0] = accb;
                sm.neg16[I0] = negb;
                sm.ndraw[I0] = ndraw;
                sm.exh[I0] = exhausted;
            }
            if (NB == 2 && (lane >> 4) == I0) sm.negv[lane] = negv;
        }
    }
    LU4_STAMP(320 + 8 * J + 6);
}

template <int NB, bool FULL, bool PRO>
__device__ __forceinline__ void lu4_block(int w, int n, const double *__restrict__ Gall, long strideG,
                                          int8_t *__restrict__ conf_slice, long conf_stride, int site0, int nsites,
                                          double *__restrict__ img_all, const SweepConsts &sc, WalkerRng *rngs,
                                          DevStats *stats, int check_sign, int *errflag, int site0p,
                                          const double *__restrict__ imgp_all)
{
    Lu4Smem<NB> &sm = *reinterpret_cast<Lu4Smem<NB> *>(lu4_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int8_t *__restrict__ cw = conf_slice + (long)w * conf_stride;
    const int myc = (FULL || lane < nsites) ? (int)cw[site0 + (FULL ? lane : min(lane, nsites - 1))] : 1;
    const unsigned long long cbits = __ballot(myc > 0);
    const WalkerRng rs = rngs[w];
    double uvec;  // lane k: the k-th uniform this chunk consumes, whichever site consumes it (DQMC.jl:573)
    {
        const unsigned long long d = rs.draw + (unsigned long long)lane;
        uvec = rs.uniforms ? (d < rs.n_uniforms ? rs.uniforms[d] : 2.0) : philox_uniform(rs.seed, d);
    }
    if (tid < 64) {
        sm.step[tid] = 0;
        for (int b = 0; b < NB; ++b) sm.xs[b][tid] = 0.0;
    }
    if (tid < 12) sm.sflag[tid] = 0;
#ifdef LU4_STAMPS
    if (w == 0 && tid == 0 && lu4_stamp_ptr) lu4_stamp_ptr[360] = (long long)__builtin_amdgcn_s_memtime();
#endif
    if (tid < 4) { sm.acc16[tid] = 0u; sm.neg16[tid] = 0u; sm.ndraw[tid] = 0; sm.exh[tid] = 0; }
    if (tid == 0) sm.abort = 0;
    if (PRO && NB == 1 && tid < 8)  // flags of the EARLY prologue (lu4_wave)
        reinterpret_cast<int *>(lu4_lds + (sizeof(Lu4Smem<NB>) + 15) / 16 * 2 + LU_STRIDE + 1536 + 1024)[tid] = 0;
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    if (wv == 0) lu4_wave<NB, FULL, PRO, 0>(n, Gall, strideG, w, site0, nsites, img_all, sc, cbits, uvec, check_sign, site0p, imgp_all);
    else if (wv == 1) lu4_wave<NB, FULL, PRO, 1>(n, Gall, strideG, w, site0, nsites, img_all, sc, cbits, uvec, check_sign, site0p, imgp_all);
    else if (wv == 2) lu4_wave<NB, FULL, PRO, 2>(n, Gall, strideG, w, site0, nsites, img_all, sc, cbits, uvec, check_sign, site0p, imgp_all);
    else lu4_wave<NB, FULL, PRO, 3>(n, Gall, strideG, w, site0, nsites, img_all, sc, cbits, uvec, check_sign, site0p, imgp_all);
    __syncthreads();
    const int nblk = (nsites + 15) / 16;
    if (!FULL) {  // blocks past the end of a short chunk: identity triangles, x = 0 (wave I0 fills block I0)
        const int I0 = wv, g = lane >> 4, ci = lane & 15;
        if (I0 >= nblk)
            for (int b = 0; b < NB; ++b) {
                const gdp img = (gdp)(img_all + (long)(w * NB + b) * LU_STRIDE);
                for (int r = 0; r < 4; ++r) {
                    const double idv = (4 * r + g == ci) ? 1.0 : 0.0;
                    for (int Jc = I0 + 1; Jc < 4; ++Jc) {
                        img[LU_OFF_U + lu_pair(I0, Jc) * LU_TILE + r * 64 + lane] = 0.0;
                        img[LU_OFF_L + lu_pair(I0, Jc) * LU_TILE + r * 64 + lane] = 0.0;
                    }
                    img[LU_OFF_PT + I0 * LU_TILE + r * 64 + lane] = idv;
                    img[LU_OFF_Q + I0 * LU_TILE + r * 64 + lane] = idv;
                }
                if (lane < 16) img[LU_IMG + 16 * I0 + lane] = 0.0;
            }
    }
    if (wv == 0) {
        const unsigned long long accbits = (unsigned long long)sm.acc16[0] | ((unsigned long long)sm.acc16[1] << 16) |
                                           ((unsigned long long)sm.acc16[2] << 32) | ((unsigned long long)sm.acc16[3] << 48);
        if (lane < nsites && ((accbits >> lane) & 1ull)) cw[site0 + lane] = (int8_t)(-myc);
        if (lane == 0) {
            rngs[w].draw = rs.draw + (unsigned long long)sm.ndraw[nblk - 1];
            if (sm.exh[nblk - 1]) rngs[w].exhausted = 1;
            stats[w].prop_local += nsites;
            stats[w].acc_local += __popcll(accbits);
            if (sm.abort) atomicOr(errflag, 2);
#ifdef LU4_STAMPS
            if (w == 0 && lu4_stamp_ptr) lu4_stamp_ptr[361] = (long long)__builtin_amdgcn_s_memtime();
#endif
            if (NB == 2)  // sign-problem statistics in site order (DQMC.jl:560-566)
                for (int s = 0; s < nsites; ++s)
                    if ((sm.neg16[s >> 4] >> (s & 15)) & 1u) magstats_push(stats[w].negative_probability, sm.negv[s]);
        }
    }
}

template <int NB, bool FULL>
__global__ __launch_bounds__(256) void sweep_lu4_kernel(int n, const double *__restrict__ Gall, long strideG,
                                                       int8_t *__restrict__ conf_slice, long conf_stride, int site0,
                                                       int nsites, double *__restrict__ img_all, SweepConsts sc,
                                                       WalkerRng *rngs, DevStats *stats, int check_sign, int *errflag)
{
    lu4_block<NB, FULL, false>(blockIdx.x, n, Gall, strideG, conf_slice, conf_stride, site0, nsites, img_all, sc, rngs,
                               stats, check_sign, errflag, 0, nullptr);
}

hipError_t launch_sweep_lu(int n, int nb, int n_walkers, const double *G, long strideG, int8_t *conf_slice,
                           long conf_stride, int site0, int nsites, double *img, SweepConsts sc, WalkerRng *rng,
                           DevStats *stats, int check_sign, int *errflag, hipStream_t s, hipEvent_t start,
                           hipEvent_t stop)
{
    if (nsites < 1 || nsites > 64 || site0 < 0 || site0 + nsites > n || nb < 1 || nb > 2) return hipErrorInvalidValue;
    const bool full = nsites == 64;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device (function attributes are per device)
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)sweep_lu4_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lu4Smem<1>));
        (void)hipFuncSetAttribute((const void *)sweep_lu4_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lu4Smem<1>));
        (void)hipFuncSetAttribute((const void *)sweep_lu4_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lu4Smem<2>));
        (void)hipFuncSetAttribute((const void *)sweep_lu4_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lu4Smem<2>));
        attr_mask |= 1u << dev;
    }
    dim3 grid(n_walkers), block(256);
#define LU4_LAUNCH(NBV, FL)                                                                                          \
    hipExtLaunchKernelGGL((sweep_lu4_kernel<NBV, FL>), grid, block, sizeof(Lu4Smem<NBV>), s, start, stop, 0, n, G, strideG, \
                          conf_slice, conf_stride, site0, nsites, img, sc, rng, stats, check_sign, errflag)
    if (nb == 1) { if (full) LU4_LAUNCH(1, true); else LU4_LAUNCH(1, false); }
    else { if (full) LU4_LAUNCH(2, true); else LU4_LAUNCH(2, false); }
#undef LU4_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// G_out[t][t'] = G_in[t][t'] + sum_s T'[s][t] G_in[site0 + s][t'] for one 64 x 64 tile; wave w owns the 16 rows
// t = m0 + 16 w .. and computes T' for them (two block-triangular solves, see the file header).
constexpr int FL_LDR = 66;  // row stride of the R0 tile in LDS (doubles): ci * 66 + g hits 32 distinct 8-byte slots
// NT = number of 16-wide column tiles per wave: the workgroup covers 64 rows x 16 NT columns (the triangular solves
// of a row tile are repeated by every workgroup of that row, so wider is cheaper: NT = 8 when n % 128 == 0).
// NCP = column passes of one workgroup (it then covers 64 rows x 16 NT NCP columns with ONE pair of triangular solves):
// used by the fused launch, where fewer, longer flush workgroups leave CUs for the elimination workgroups
template <bool FULL, int NT, int NCP = 1>
__device__ __forceinline__ void flush_lu_body(int bid, int n, int n_units, const double *__restrict__ Gin_all,
                                              double *__restrict__ Gout_all, long strideG, int site0, int nsites,
                                              const double *__restrict__ img_all, int tiles_m, int tiles_n,
                                              int ncp_rt = 1)  // NCP == 0: the pass count is this run-time value
{
    const int ncp = NCP > 0 ? NCP : ncp_rt;
    double *fsm = lu4_lds;
    double *img = fsm;                 // [LU_IMG + 64]
    double *xs = fsm + LU_IMG;
    double *Rl = fsm + LU_STRIDE;      // [16 NT t'][FL_LDR]: R0[s][t'] at Rl[t' * FL_LDR + s]
    const int xcd = bid & 7, seq = bid >> 3;
    const int T = tiles_m * tiles_n;
    const int unit = (seq / T) * 8 + xcd;
    if (unit >= n_units) return;
    const int tile = seq % T;
    const int m0 = (tile % tiles_m) * 64, n00 = (tile / tiles_m) * (16 * NT * ncp);
    int n0 = n00;
    const gcdp Gin = (gcdp)(Gin_all + (long)unit * strideG);
    const gdp Gout = (gdp)(Gout_all + (long)unit * strideG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ci = lane & 15;
    const int t = m0 + 16 * wave + ci;             // my row of G (contiguous direction)
    const int tq = FULL ? t : min(t, n - 1);

    FL_STAMP(0);
    // the tile of G itself first (the kernel is bound by this read-modify-write)
    d4 acc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tp = n0 + 16 * jt + 4 * r + g;
            acc[jt][r] = Gin[tq + (long)n * (FULL ? tp : min(tp, n - 1))];
        }
    // C0'[s][t] = G_in[t][site0 + s] - delta
    d4 az[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 16 * J + 4 * r + g;
            const int col = site0 + (FULL ? s : min(s, nsites - 1));
            double v = Gin[tq + (long)n * col];
            if (!FULL && (s >= nsites || t >= n)) v = 0.0;
            az[J][r] = v - ((t == site0 + s) ? 1.0 : 0.0);
        }
    // static operands of the triangles and x
    {
        const gcd2p src = reinterpret_cast<gcd2p>((gcdp)(img_all + (long)unit * LU_STRIDE));
        d2v *dst = reinterpret_cast<d2v *>(img);
        for (int i = tid; i < LU_STRIDE / 2; i += 256) dst[i] = src[i];
    }
    // R0 tile: thread -> column t' = pass * 64 + tid / 4, 16 consecutive s
#pragma unroll
    for (int pass = 0; pass < NT / 4; ++pass) {
        const int tl = pass * 64 + (tid >> 2), tp = n0 + tl, s0 = (tid & 3) * 16;
        double *d = Rl + tl * FL_LDR + s0;
        if (FULL) {
            // (a wave's request covers 2 columns x 512 contiguous bytes = 8 cache lines; one column per 4 lanes with 16
            // doubles each touched 64 lines per request)
            d2v rv[8];  // (requests first, LDS stores after: written as load-store pairs they are waited for one by one)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = i * 256 + tid, col = pass * 64 + (pc >> 5), w2 = 2 * (pc & 31);
                rv[i] = *reinterpret_cast<gcd2p>(Gin + (long)n * (n0 + col) + site0 + w2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = i * 256 + tid, col = pass * 64 + (pc >> 5), w2 = 2 * (pc & 31);
                *reinterpret_cast<d2v *>(Rl + col * FL_LDR + w2) = rv[i];
            }
        } else {
            for (int i = 0; i < 16; ++i)
                d[i] = (tp < n && s0 + i < nsites) ? Gin[(long)n * tp + site0 + s0 + i] : 0.0;
        }
    }
    __syncthreads();
    FL_STAMP(1);

    // Z_J = PT_J (C0'_J + sum_{K<J} Uu_KJ' X_K Z_K);  XZ_J = X_J Z_J
    d4 xz[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int K = 0; K < J; ++K)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                az[J] = MFMA(img[LU_OFF_U + lu_pair(K, J) * LU_TILE + q * 64 + lane], xz[K][q], az[J]);
        d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) z = MFMA(img[LU_OFF_PT + J * LU_TILE + q * 64 + lane], az[J][q], z);
#pragma unroll
        for (int r = 0; r < 4; ++r) xz[J][r] = z[r] * xs[16 * J + 4 * r + g];
    }
    // T'_J = Q_J' (XZ_J + X_J sum_{K>J} L_KJ' T'_K)
    d4 tt[4];
#pragma unroll
    for (int J = 3; J >= 0; --J) {
        d4 a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int K = J + 1; K < 4; ++K)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a = MFMA(img[LU_OFF_L + lu_pair(J, K) * LU_TILE + q * 64 + lane], tt[K][q], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = xz[J][r] + xs[16 * J + 4 * r + g] * a[r];
        d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) o = MFMA(img[LU_OFF_Q + J * LU_TILE + q * 64 + lane], a[q], o);
        tt[J] = o;
    }
    if constexpr (NCP == 0 && FULL && NT == 4) {
        // run-time pass count (throughput regime: more workgroups than one round): software pipeline over the passes.
        // While the MFMAs of pass cp run, the G tile and the R0 tile of pass cp + 1 are already on their way (registers);
        // the R0 tiles alternate between the buffer behind the solves' operands and - the solves being done - the
        // operands' own place, so a pass needs ONE barrier and LDS stays at 75 KB (two workgroups per CU).
        const int rb_off[2] = {(int)(Rl - fsm), 0};  // offsets, not pointers: a run-time choice between two pointers loses
                                                     // the LDS address space (every operand read became a flat load)
        FL_STAMP(2);
        __syncthreads();  // every wave is done with the solves' operands (fsm is reused from pass 1 on)
        FL_STAMP(3);
        for (int cp = 0; cp < ncp; ++cp) {
            const bool more = cp + 1 < ncp;
            const int n1 = n00 + (more ? cp + 1 : cp) * 64;  // (last pass: harmless re-read of its own tile)
            d4 nxt[NT];
            double r0n[16];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) nxt[jt][r] = Gin[t + (long)n * (n1 + 16 * jt + 4 * r + g)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = i * 256 + tid;
                const d2v v = *reinterpret_cast<gcd2p>(Gin + (long)n * (n1 + (pc >> 5)) + site0 + 2 * (pc & 31));
                r0n[2 * i] = v.x;
                r0n[2 * i + 1] = v.y;
            }
            const double *Rc = fsm + rb_off[cp & 1];
#pragma unroll
            for (int K = 0; K < 4; ++K)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sk = 16 * K + 4 * q + g;
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) acc[jt] = MFMA(Rc[(16 * jt + ci) * FL_LDR + sk], tt[K][q], acc[jt]);
                }
            n0 = n00 + cp * 64;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gout[t + (long)n * (n0 + 16 * jt + 4 * r + g)] = acc[jt][r];
            if (more) {
                // (the other buffer was last read in pass cp - 1; the barrier of that pass lies between)
                double *nb = fsm + rb_off[(cp + 1) & 1];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pc = i * 256 + tid;
                    *reinterpret_cast<double2 *>(nb + (pc >> 5) * FL_LDR + 2 * (pc & 31)) = make_double2(r0n[2 * i], r0n[2 * i + 1]);
                }
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) acc[jt] = nxt[jt];
                __syncthreads();
            }
            FL_STAMP(4 + cp);
        }
        return;
    }
    for (int cp = 0; cp < ncp; ++cp) {
        if (cp > 0) {  // next 16 NT columns: G tile and R0 tile again (the solves are done)
            n0 = n00 + cp * 16 * NT;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tp = n0 + 16 * jt + 4 * r + g;
                    acc[jt][r] = Gin[tq + (long)n * (FULL ? tp : min(tp, n - 1))];
                }
            __syncthreads();  // everybody is done with the previous R0 tile
#pragma unroll
            for (int pass = 0; pass < NT / 4; ++pass) {
                const int tl = pass * 64 + (tid >> 2), tp = n0 + tl, s0 = (tid & 3) * 16;
                double *d = Rl + tl * FL_LDR + s0;
                if (FULL) {
                    d2v rv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int pc = i * 256 + tid, col = pass * 64 + (pc >> 5), w2 = 2 * (pc & 31);
                        rv[i] = *reinterpret_cast<gcd2p>(Gin + (long)n * (n0 + col) + site0 + w2);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int pc = i * 256 + tid, col = pass * 64 + (pc >> 5), w2 = 2 * (pc & 31);
                        *reinterpret_cast<d2v *>(Rl + col * FL_LDR + w2) = rv[i];
                    }
                } else {
                    for (int i = 0; i < 16; ++i)
                        d[i] = (tp < n && s0 + i < nsites) ? Gin[(long)n * tp + site0 + s0 + i] : 0.0;
                }
            }
            __syncthreads();
        }
        // G tile += T R0
#pragma unroll
        for (int K = 0; K < 4; ++K)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sk = 16 * K + 4 * q + g;
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
                    acc[jt] = MFMA(Rl[(16 * jt + ci) * FL_LDR + sk], tt[K][q], acc[jt]);
            }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tp = n0 + 16 * jt + 4 * r + g;
                if (FULL || (t < n && tp < n)) Gout[t + (long)n * tp] = acc[jt][r];
            }
    }
}

template <bool FULL, int NT, int NCP = 1>
__global__ __launch_bounds__(256) void sweep_flush_lu_kernel(int n, int n_units, const double *__restrict__ Gin_all,
                                                            double *__restrict__ Gout_all, long strideG, int site0,
                                                            int nsites, const double *__restrict__ img_all,
                                                            int tiles_m, int tiles_n, int ncp_rt)
{
    flush_lu_body<FULL, NT, NCP>(blockIdx.x, n, n_units, Gin_all, Gout_all, strideG, site0, nsites, img_all, tiles_m, tiles_n,
                                 ncp_rt);
}

// One launch per chunk boundary: the first n_walkers workgroups eliminate chunk `site0` (adding the not yet applied
// previous chunk to their 64 x 64 block themselves), all other workgroups apply the previous chunk to G out of place.
struct SweepFusedArgs {
    int n, n_walkers, n_units;
    const double *Gin;
    double *Gout;
    long strideG;
    int8_t *conf_slice;
    long conf_stride;
    int site0, site0p;
    double *img;         // written by the elimination of this chunk
    const double *imgp;  // images of the previous chunk
    SweepConsts sc;
    WalkerRng *rngs;
    DevStats *stats;
    int check_sign;
    int *errflag;
    int tiles_m, tiles_n;
};
template <int NB, int NT, int NCP>
__global__ __launch_bounds__(256) void sweep_fused_kernel(SweepFusedArgs a)
{
    if ((int)blockIdx.x < a.n_walkers)
        lu4_block<NB, true, true>(blockIdx.x, a.n, a.Gin, a.strideG, a.conf_slice, a.conf_stride, a.site0, 64, a.img, a.sc,
                                  a.rngs, a.stats, a.check_sign, a.errflag, a.site0p, a.imgp);
    else
        flush_lu_body<true, NT, NCP>((int)blockIdx.x - a.n_walkers, a.n, a.n_units, a.Gin, a.Gout, a.strideG, a.site0p, 64,
                                     a.imgp, a.tiles_m, a.tiles_n);
}


hipError_t launch_sweep_flush_lu(int n, int n_units, const double *Gin, double *Gout, long strideG, int site0,
                                 int nsites, const double *img, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    const bool wide = n % 128 == 0;
    const int nt = wide ? 8 : 4;
    const int tm = (n + 63) / 64, tn = (n + 16 * nt - 1) / (16 * nt);
    const int groups = (n_units + 7) / 8;
    const size_t lds = ((size_t)LU_STRIDE + 16 * nt * FL_LDR) * sizeof(double);
    const bool full = (n % 64 == 0) && nsites == 64;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device (function attributes are per device)
    if (!(attr_mask & (1u << dev))) {
        const int big = (int)(((size_t)LU_STRIDE + 128 * FL_LDR) * sizeof(double));
        (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        attr_mask |= 1u << dev;
    }
#define FL_LAUNCH(FU, NTV)                                                                                          \
    hipExtLaunchKernelGGL((sweep_flush_lu_kernel<FU, NTV>), dim3(groups * 8 * tm * tn), dim3(256), lds, s, start, stop, 0, \
                          n, n_units, Gin, Gout, strideG, site0, nsites, img, tm, tn, 1)
    // two column passes per workgroup (one pair of triangular solves per 64-row tile instead of two) when the grid is
    // more than one round of workgroups anyway: 512 units (config 4 on one GPU) 293 -> 230 us; a single round
    // (32 units: 256 workgroups) is faster with one pass each (16.7 vs 24 us).  DQMC_FLUSH_NCP2 forces it.
    const bool ncp2_env = kernel_switches().flush_ncp2;
    static int n_cus[32] = {0};
    if (dev >= 0 && dev < 32 && n_cus[dev] == 0) {
        hipDeviceProp_t prop;
        n_cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    const bool ncp2 = ncp2_env || (dev >= 0 && dev < 32 && groups * 8 * tm * tn > n_cus[dev]);
    // (n % 256 == 0 used to take 128-column passes here: 109 KB of LDS, one workgroup per CU, 230 us per launch of 512 units;
    // 64-column passes - 75 KB, two workgroups per CU - 190 us, software-pipelined 187.  A third workgroup per CU (R0 tile
    // aliased onto the solves' operands, 41 KB): 193 us; a late start of every other workgroup: slower by the delay.
    // In-kernel stamps (tools/fl_stamps.py): 38 k cycles until the operands of a workgroup are there, 8 - 10 k of solves,
    // 12 - 16 k per pass of 4 k cycles of MFMA - all resident workgroups fetch at ~17 B/cycle/CU, which is what the part
    // delivers to every CU at once; PMC: HBM traffic = algorithmic (profiles/r03_pmc_flush_512units.txt))
    if (ncp2 && wide && full && n % 256 == 0 && ncp2_env) {  // (the former form: DQMC_FLUSH_NCP2, for A/B and under test)
        static unsigned m2 = 0;
        if (!(m2 & (1u << dev))) {
            (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<true, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(((size_t)LU_STRIDE + 128 * FL_LDR) * sizeof(double)));
            m2 |= 1u << dev;
        }
        hipExtLaunchKernelGGL((sweep_flush_lu_kernel<true, 8, 2>), dim3(groups * 8 * tm * (tn / 2)), dim3(256), lds, s, start, stop,
                              0, n, n_units, Gin, Gout, strideG, site0, nsites, img, tm, tn / 2, 2);
        return hipGetLastError();
    }
    // n not a multiple of 128 (64-column tiles, e.g. n = 576: 9 x 9 tiles per unit, every row tile's solves done 9
    // times over) and more than one round of workgroups: ONE workgroup per 64-row tile, all column tiles in passes
    if (ncp2 && full && (n + 63) / 64 > 1) {
        const int tn4 = (n + 63) / 64;
        const size_t lds4 = ((size_t)LU_STRIDE + 64 * FL_LDR) * sizeof(double);
        static unsigned m0 = 0;
        if (!(m0 & (1u << dev))) {
            (void)hipFuncSetAttribute((const void *)sweep_flush_lu_kernel<true, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(((size_t)LU_STRIDE + 128 * FL_LDR) * sizeof(double)));
            m0 |= 1u << dev;
        }
        hipExtLaunchKernelGGL((sweep_flush_lu_kernel<true, 4, 0>), dim3(groups * 8 * tm), dim3(256), lds4, s, start, stop, 0, n,
                              n_units, Gin, Gout, strideG, site0, nsites, img, tm, 1, tn4);
        return hipGetLastError();
    }
    if (wide) { if (full) FL_LAUNCH(true, 8); else FL_LAUNCH(false, 8); }
    else { if (full) FL_LAUNCH(true, 4); else FL_LAUNCH(false, 4); }
#undef FL_LAUNCH
    return hipGetLastError();
}

hipError_t launch_sweep_fused(int n, int nb, int n_walkers, const double *Gin, double *Gout, long strideG,
                              int8_t *conf_slice, long conf_stride, int site0, int site0p, double *img,
                              const double *imgp, SweepConsts sc, WalkerRng *rng, DevStats *stats, int check_sign,
                              int *errflag, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    if (n % 64 != 0 || site0 % 64 != 0 || site0p % 64 != 0 || site0 + 64 > n || site0p + 64 > n || nb < 1 || nb > 2)
        return hipErrorInvalidValue;
    const int n_units = n_walkers * nb;
    const bool wide = n % 128 == 0;
    const int nt = wide ? 8 : 4;
    SweepFusedArgs a;
    a.n = n; a.n_walkers = n_walkers; a.n_units = n_units; a.Gin = Gin; a.Gout = Gout; a.strideG = strideG;
    a.conf_slice = conf_slice; a.conf_stride = conf_stride; a.site0 = site0; a.site0p = site0p; a.img = img; a.imgp = imgp;
    a.sc = sc; a.rngs = rng; a.stats = stats; a.check_sign = check_sign; a.errflag = errflag;
    // two column passes per flush workgroup when that makes the whole launch co-resident (one workgroup per CU)
    const int ncp = (wide && n % 256 == 0) ? 2 : 1;
    a.tiles_m = n / 64; a.tiles_n = n / (16 * nt * ncp);
    const int groups = (n_units + 7) / 8;
    const int flush_blocks = groups * 8 * a.tiles_m * a.tiles_n;
    const size_t lds_flush = ((size_t)LU_STRIDE + 16 * nt * FL_LDR) * sizeof(double);
    const size_t lds_lu = nb == 1 ? sizeof(Lu4Smem<1>) : sizeof(Lu4Smem<2>);
    // NB == 1: the prologue stages the previous chunk's images behind the shared structure (lu4_wave, STAGE)
    // + deltas (1536 doubles), wave 0's T' tiles (1024), flags (EARLY prologue)
    const size_t lds_pro = nb == 1 ? (sizeof(Lu4Smem<1>) + 15) / 16 * 16 + (LU_STRIDE + 1536 + 1024 + 8) * sizeof(double) : lds_lu;
    const size_t lds_e = lds_pro > lds_lu ? lds_pro : lds_lu;
    const size_t lds = lds_flush > lds_e ? lds_flush : lds_e;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<1, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<1, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<1, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<2, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<2, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)sweep_fused_kernel<2, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_mask |= 1u << dev;
    }
    dim3 grid(n_walkers + flush_blocks), block(256);
#define FU_LAUNCH(NBV, NTV, NCPV) hipExtLaunchKernelGGL((sweep_fused_kernel<NBV, NTV, NCPV>), grid, block, lds, s, start, stop, 0, a)
    if (nb == 1) { if (ncp == 2) FU_LAUNCH(1, 8, 2); else if (wide) FU_LAUNCH(1, 8, 1); else FU_LAUNCH(1, 4, 1); }
    else { if (ncp == 2) FU_LAUNCH(2, 8, 2); else if (wide) FU_LAUNCH(2, 8, 1); else FU_LAUNCH(2, 4, 1); }
#undef FU_LAUNCH
    return hipGetLastError();
}

#ifdef LU4_STAMPS
extern "C" int dqmc_debug_lu4_stamps(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(lu4_stamp_ptr), &devptr, sizeof(void *));
}
#endif

}  // namespace dqmc
