// slab.hip — chains of left products with the column slab resident in LDS (n = 256).
//
//   X_s = post_s (.) ( A_s * ( pre_s (.) X_{s-1} ) ),  s = 1 .. nsteps,    out = X_nsteps (.) colscale
//
// for the sequences of slice-matrix products that the sweep runs back to back on one matrix:
//   * add_slice_sequence_left/right (stack.jl:272-311): s = safe_mult products B_l X or B_l' X
//     (slice_matrices.jl:42-48, 70-76) -- one launch instead of ten GEMM launches;
//   * wrap_greens! (stack.jl:491-500): G <- B G B^-1 = eT2 (eV (G (eV^-1 eTinv2))) evaluated column slab by column slab
//     (X_0 = a slab of the CONSTANT eTinv2, A_1 = G, A_2 = eT2) -- one launch instead of two.
//
// One workgroup (4 waves) owns a 256 x 32 column slab of X: 8 workgroups per unit, 256 workgroups for 32 units, mapped
// so that the 8 slabs of a unit sit on one XCD (they stream the same A through that XCD's L2).  The slab lives in LDS
// (two buffers of 32 x 258 doubles: a step reads one and writes the other, ONE LDS-only barrier per step); A_s is
// never staged: no two waves of a workgroup share an A element (wave w owns rows 64 w .. 64 w + 63 of the product), so
// every lane loads its MFMA A operand straight from L2 into registers, four k-pairs ahead, also across the step
// boundary.  v_mfma_f64_16x16x4_f64, 4 x 2 tiles per wave; the k index of the two MFMAs of a pair is interleaved
// (k = 8 p + 2 g and 8 p + 2 g + 1) so that one ds_read_b128 feeds two MFMAs, and the tile rows are permuted
// (tile row 4 r + g <-> matrix row 4 g + r) so that a lane's four accumulator registers are four consecutive rows:
// slab write-back and the final store are 32-byte pieces, the final store covers whole 128-byte lines.
// Scalings (exp(+-lambda conf), HubbardModelAttractive.jl:100-110 / HubbardModelRepulsive.jl:113-126) are applied to
// the accumulators at write-back: post_s and pre_{s+1} together, once per element.
#include "kernels.h"
#include <hip/hip_ext.h>

namespace dqmc {

typedef double d4 __attribute__((ext_vector_type(4)));
#define SLAB_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define SLAB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int SL_N = 256;
constexpr int SL_W = 32;     // slab width
constexpr int SL_LD = 258;   // LDS column stride (doubles): 16 lanes x 16 B of one read pass hit 64 distinct banks
constexpr int SL_RING = 4;   // A operands in flight: pairs p .. p + 3

// exp(sign lambda conf) of row i (block b): vs_conf() of engine.cpp
__device__ __forceinline__ double slab_conf_val(const int8_t *conf, int i, int sign, int blk, double epl, double eml)
{
    const bool plus = (conf[i] > 0) == (sign > 0);
    return (plus != (blk != 0)) ? epl : eml;
}

#ifdef SLAB_STAMPS  // diagnostic build only (tools/slab_stamps.py): shader-clock / 100 MHz stamps of one workgroup
__device__ long long *slab_stamp_ptr = nullptr;
#ifndef SLAB_STAMP_BID
#define SLAB_STAMP_BID 0
#endif
#define SLAB_STAMP(k)                                                                                   \
    do {                                                                                                \
        if (blockIdx.x == SLAB_STAMP_BID && (threadIdx.x & 63) == 0 && slab_stamp_ptr) {                \
            slab_stamp_ptr[64 * (threadIdx.x >> 6) + 2 * (k)] = (long long)__builtin_amdgcn_s_memtime(); \
            slab_stamp_ptr[64 * (threadIdx.x >> 6) + 2 * (k) + 1] = (long long)__builtin_amdgcn_s_memrealtime(); \
        }                                                                                               \
    } while (0)
#else
#define SLAB_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256) void slab_chain_kernel(SlabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double slab_lds[];
    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int unit = (seq >> 3) * 8 + xcd, slab = seq & 7;
    if (unit >= a.n_units) return;
    SLAB_STAMP(0);
    // (nb is 1 or 2 - BlockDiagonal has two blocks, blockdiagonal.jl:13-36: a shift instead of the ~25 instructions of an
    // integer division in front of the first requests)
    const int wk = a.nb == 2 ? unit >> 1 : unit, blk = a.nb == 2 ? unit & 1 : 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, ci = lane & 15;
    double *buf0 = slab_lds, *buf1 = slab_lds + SL_W * SL_LD;
    const long conf_off = (long)wk * a.conf_stride;

    // my A rows: tile row ci <-> matrix row 4 (ci & 3) + (ci >> 2) of each 16-row tile; k = 8 p + 2 g (+ 1)
    const int arow = 64 * w + 4 * (ci & 3) + (ci >> 2);
    auto a_ptr = [&](int s) {
        return a.st[s].A + (long)unit * a.st[s].su + (long)blk * a.st[s].sb + arow + (long)SL_N * (2 * g);
    };
    double areg[SL_RING][4][2];  // [ring slot][row tile][k of the pair]
    // request of one pair of k columns (8 loads): no condition around it and no branch in the k-loop, see below
    auto request = [&](int slot, const double *p) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            areg[slot][rt][0] = p[16 * rt];
            areg[slot][rt][1] = p[16 * rt + SL_N];
        }
    };
    double csl[2] = {1.0, 1.0};  // column scale of the final store (stack.jl:281 / :305, or eV of the wrap)
    // ---- X_0 slab -> LDS, scaled by pre_1.  All sixteen requests of a thread (and the 32 HS-field bytes of its rows, as
    // four 8-byte words) are in flight together: written as load - scale - store per element, each element waited for
    // its own trip to the L2 (the field byte sat behind a branch): sixteen trips before the first product could start.
    {
        const double *X0 = a.X0 + (long)unit * a.x_su + (long)blk * a.x_sb + (long)SL_N * (SL_W * slab);
        const int col = tid >> 3, r0 = 32 * (tid & 7);
        const int8_t *pc = a.st[0].pre_conf ? a.st[0].pre_conf + conf_off : nullptr;
        typedef double d2x __attribute__((ext_vector_type(2)));
        d2x xv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xv[i] = *reinterpret_cast<const d2x *>(X0 + (long)SL_N * col + r0 + 2 * i);
        // (rows r0 .. r0 + 31 of the field: r0 is a multiple of 32 and a slice starts on a multiple of SL_N bytes)
        const unsigned long long *pw = reinterpret_cast<const unsigned long long *>(pc ? pc + r0 : reinterpret_cast<const int8_t *>(X0));
        unsigned long long cw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cw[i] = pw[i];
        // (the column scale of the final store travels with the slab, raw and without a branch: one wait for everything)
        const int gc0 = SL_W * slab + ci;
        const double *cdp = a.col_d ? a.col_d + (long)unit * a.col_stride + gc0 : X0;
        const int8_t *ccp = a.col_conf ? a.col_conf + conf_off + gc0 : reinterpret_cast<const int8_t *>(X0);
        const double cd0 = cdp[0], cd1 = cdp[16];
        const int8_t cc0 = ccp[0], cc1 = ccp[16];
        {   // first A operands: in flight before X_0 is staged, but asked for BEHIND the slab - the wrap's A_1 = G comes from
            // HBM, and requests retire in order: ahead of the slab they held up its (cache-resident) data
            const double *ap0 = a_ptr(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < SL_RING - 1; ++i) request(i, ap0 + (long)SL_N * 8 * i);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
        SLAB_STAMP(20);  // all prologue requests issued
#ifdef SLAB_STAMPS
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        SLAB_STAMP(21);  // slab data arrived (the A operands may still be in flight)
#endif
        const bool sp = a.st[0].pre_sign > 0, bn = blk != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double fx = 1.0, fy = 1.0;
            if (pc) {
                const int8_t b0 = (int8_t)(cw[(2 * i) >> 3] >> (8 * ((2 * i) & 7)));
                const int8_t b1 = (int8_t)(cw[(2 * i + 1) >> 3] >> (8 * ((2 * i + 1) & 7)));
                fx = (((b0 > 0) == sp) != bn) ? a.epl : a.eml;
                fy = (((b1 > 0) == sp) != bn) ? a.epl : a.eml;
            }
            d2x v = xv[i];
            v.x *= fx;
            v.y *= fy;
            *reinterpret_cast<d2x *>(buf0 + col * SL_LD + r0 + 2 * i) = v;
        }
        SLAB_STAMP(22);  // slab scaled and written to LDS
        if (a.col_d) {
            csl[0] = cd0;
            csl[1] = cd1;
        } else if (a.col_conf) {
            csl[0] = ((((cc0 > 0) == (a.col_sign > 0)) != bn)) ? a.epl : a.eml;
            csl[1] = ((((cc1 > 0) == (a.col_sign > 0)) != bn)) ? a.epl : a.eml;
        }
    }
    __syncthreads();
    SLAB_STAMP(1);

    d4 acc[4][2];
    for (int s = 0; s < a.nsteps; ++s) {
        const double *src = (s & 1) ? buf1 : buf0;
        double *dst = (s & 1) ? buf0 : buf1;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (d4){0.0, 0.0, 0.0, 0.0};
        // The scalings of this step's write-back are requested HERE, ahead of the 512 MFMAs, as one 32-bit word per row tile
        // (rows 64 w + 16 rt + 4 g .. + 3 of the HS field; a slice starts on a multiple of SL_N bytes): asked for at the
        // write-back they were eight dependent trips to the L2 per step (post, wait, pre, wait per row tile; each wait also
        // drained the A operands already in flight for the next step) - 6 of the 19.8 us a step took.
        const bool last = s + 1 == a.nsteps;
        const int8_t *c_post = a.st[s].post_conf ? a.st[s].post_conf + conf_off : nullptr;
        const int8_t *c_pre = (!last && a.st[s + 1].pre_conf) ? a.st[s + 1].pre_conf + conf_off : nullptr;
        const int s_post = a.st[s].post_sign, s_pre = last ? 0 : a.st[s + 1].pre_sign;
        unsigned wpost[4] = {0u, 0u, 0u, 0u}, wpre[4] = {0u, 0u, 0u, 0u};
        if (c_post) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) wpost[rt] = *reinterpret_cast<const unsigned *>(c_post + 64 * w + 16 * rt + 4 * g);
        }
        if (c_pre) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) wpre[rt] = *reinterpret_cast<const unsigned *>(c_pre + 64 * w + 16 * rt + 4 * g);
        }
        const double *bp = src + ci * SL_LD + 2 * g;  // + 16 ct SL_LD + 8 p
        double2 breg[2][2];                            // [parity of p][ct]
        breg[0][0] = *reinterpret_cast<const double2 *>(bp);
        breg[0][1] = *reinterpret_cast<const double2 *>(bp + 16 * SL_LD);
        // The k-loop is ONE basic block per SL_RING pairs, and its schedule is written down: in the shadow of every second
        // MFMA (64 cycles of the pipe, 4 of issue) goes one of the 8 operand requests of pair p + SL_RING - 1 or one of the
        // two LDS reads of pair p + 1.  Left to itself the compiler emitted the 16 MFMAs of a pair back to back and the ~25
        // other instructions behind them: the pipe idled ~330 of every 1 350 cycles (tools/slab_stamps.py: 43 300 cycles
        // per step against 32 768 of MFMA issue).  Hence no branch inside: the stream runs on into the next step's A (the
        // last step's own A again at the very end - asked for, never used), the B read of pair 32 wraps to pair 0.
        const double *apc = a_ptr(s), *apn = a_ptr(min(s + 1, a.nsteps - 1));
#pragma unroll 1
        for (int p0 = 0; p0 < 32; p0 += SL_RING) {  // (not unrolled further: the written schedule is per SL_RING pairs)
#pragma unroll
            for (int i = 0; i < SL_RING; ++i) {
                const int p = p0 + i, pr = p + SL_RING - 1;
                request((i + SL_RING - 1) % SL_RING, (pr < 32 ? apc : apn) + (long)SL_N * 8 * (pr & 31));
                breg[(i + 1) & 1][0] = *reinterpret_cast<const double2 *>(bp + 8 * ((p + 1) & 31));
                breg[(i + 1) & 1][1] = *reinterpret_cast<const double2 *>(bp + 16 * SL_LD + 8 * ((p + 1) & 31));
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    acc[rt][0] = SLAB_MFMA(areg[i][rt][0], breg[i & 1][0].x, acc[rt][0]);
                    acc[rt][1] = SLAB_MFMA(areg[i][rt][0], breg[i & 1][1].x, acc[rt][1]);
                }
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    acc[rt][0] = SLAB_MFMA(areg[i][rt][1], breg[i & 1][0].y, acc[rt][0]);
                    acc[rt][1] = SLAB_MFMA(areg[i][rt][1], breg[i & 1][1].y, acc[rt][1]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one operand request
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    if (k == 2 || k == 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
                }
            }
        }
        SLAB_STAMP(2 + 2 * s);
        // ---- write-back: rows 64 w + 16 rt + 4 g + r (r = 0..3), column 16 ct + ci; scale post_s (.) pre_{s+1}
        const bool bn = blk != 0;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = 64 * w + 16 * rt + 4 * g;
            double f[4] = {1.0, 1.0, 1.0, 1.0};
            if (c_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f[r] = ((((int8_t)(wpost[rt] >> (8 * r)) > 0) == (s_post > 0)) != bn) ? a.epl : a.eml;
            }
            if (c_pre) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f[r] *= ((((int8_t)(wpre[rt] >> (8 * r)) > 0) == (s_pre > 0)) != bn) ? a.epl : a.eml;
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = 16 * ct + ci;
                double v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[rt][ct][r] * f[r];
                if (!last) {
                    double *d = dst + col * SL_LD + row;
                    *reinterpret_cast<double2 *>(d) = make_double2(v[0], v[1]);
                    *reinterpret_cast<double2 *>(d + 2) = make_double2(v[2], v[3]);
                } else {
                    const int gc = SL_W * slab + col;  // global column
                    const double cs = csl[ct];
                    double *o = a.out + (long)unit * a.out_su + (long)SL_N * gc + row;
                    *reinterpret_cast<double2 *>(o) = make_double2(v[0] * cs, v[1] * cs);
                    *reinterpret_cast<double2 *>(o + 2) = make_double2(v[2] * cs, v[3] * cs);
                }
            }
        }
        if (!last) SLAB_BARRIER();
        SLAB_STAMP(3 + 2 * s);
    }
}
#ifdef SLAB_STAMPS
extern "C" int dqmc_debug_slab_stamps(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(slab_stamp_ptr), &devptr, sizeof(void *));
}
#endif

hipError_t launch_slab_chain(const SlabArgs &a, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    if (a.nsteps < 1 || a.nsteps > SLAB_MAX_STEPS || a.nb < 1 || a.nb > 2) return hipErrorInvalidValue;
    const size_t lds = 2 * SL_W * SL_LD * sizeof(double);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;  // per device
    if (!(attr_mask & (1u << dev))) {
        hipError_t e = hipFuncSetAttribute((const void *)slab_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_mask |= 1u << dev;
    }
    const int groups = (a.n_units + 7) / 8;
    const dim3 grid(groups * 64), block(256);
    if (start) hipExtLaunchKernelGGL(slab_chain_kernel, grid, block, lds, s, start, stop, 0, a);
    else hipLaunchKernelGGL(slab_chain_kernel, grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace dqmc
