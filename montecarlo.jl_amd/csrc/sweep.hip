// sweep.hip — sweep_spatial (src/flavors/DQMC/DQMC.jl:546-582) with propose_local /
// accept_local! of HubbardModelAttractive.jl:113-155 and HubbardModelRepulsive.jl:128-232,
// plus the small bookkeeping kernels (propagation-error check, measurement sums).
//
// The site loop is strictly sequential, so one workgroup owns one walker.  Accepted
// rank-1 Sherman–Morrison updates are DELAYED inside a chunk of KD sites: thread t keeps
// its row of U' (n x cnt) and its column of V (cnt x n) in registers,
//     G = G0 + U' V,   U'[:,j] = -(e_i - G[:,i]) * x,   V[j,:] = G[i,:],
// so a proposal only needs the current diagonal (kept in LDS) and an accept costs 2*cnt
// FMAs per thread instead of a 2 n^2 pass over G.  The chunk's U', V^T are written out
// zero padded and the host flushes them with one MFMA GEMM (G0 += U' V) for all walkers.
// This is a re-association of the reference's arithmetic only.
#include "kernels.h"
#include <cstdlib>

namespace dqmc {

constexpr int SW_GROUP = 8;   // sites per group (static register indices inside a group)
constexpr int SW_KD = 64;     // sites per chunk = update slots per flush

// Two-level delayed updates.  Within a chunk of KD sites the accepted Sherman-Morrison updates
// are kept as G = G0 + U'V (slot j = j-th accept of the chunk):
//   * the slot history U'[t][m], V[m][t] of thread t lives in the global arrays that the flush
//     GEMM reads anyway (own-lane coalesced stores at accept, own-lane loads at group starts);
//   * LDS keeps only what other threads need: UiT[m][s] = U'[site0+s][m], ViT[m][s] = V[m][site0+s]
//     for the chunk's own sites, the current diagonal dg[s], the uniforms of the chunk;
//   * the rows/columns of G at the 8 sites of the current GROUP are held current in registers:
//     corrected once at the group start from all older slots (batched: one history load feeds 16
//     FMAs), then updated eagerly at every accept inside the group (<= 14 FMAs).
// An accept therefore costs O(1) per thread on the critical path instead of a 2*cnt loop.
// Layout of the dynamic LDS region (doubles unless noted):
//   UiT[nb][KD][KD], ViT[nb][KD][KD], dg[2][KD], ul[KD], negv[KD], cs[KD] (int), flip[KD] (int)
// MODEL (0 attractive, 1 repulsive) and FULL are compile-time: FULL = no padding lanes (n % 64 == 0) and the
// chunk is one aligned block of 64 sites, i.e. exactly one wave per block owns the chunk's rows.  PMC counters
// show the per-site instruction stream (a third of it scalar: exec-mask branches) to be the limiter; the
// specialisation removes the per-lane predicates and the other model's code from that stream.
template <int MAXT, int MODEL, bool FULL>
__global__ __launch_bounds__(MAXT) void sweep_chunk_kernel(int n, int nb, int model_rt, double *__restrict__ Gall,
                                                          long strideG, int8_t *__restrict__ conf_slice,
                                                          long conf_stride, int site0, int nsites,
                                                          double *__restrict__ Uall, double *__restrict__ VTall,
                                                          long strideUV, SweepConsts sc, WalkerRng *rngs,
                                                          DevStats *stats, int check_sign)
{
    constexpr int KD = SW_KD;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int npad = (n + 63) & ~63;
    double *UiT = sm;
    double *ViT = UiT + (size_t)nb * KD * KD;
    double *dg = ViT + (size_t)nb * KD * KD;  // [2][KD]
    double *ul = dg + 2 * KD;
    double *negv = ul + KD;
    int *cs = (int *)(negv + KD);
    int *flip = cs + KD;

    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    const int b = tid / npad, t = tid - b * npad;  // wave-uniform block index
    (void)model_rt;
    constexpr int model = MODEL;
    const bool active = FULL ? true : t < n;
    const int unit = w * nb + b;
    const double *__restrict__ G = Gall + (long)unit * strideG;
    double *Uo = Uall + (long)unit * strideUV;   // slot m of thread t at Uo[t + n*m]
    double *VTo = VTall + (long)unit * strideUV;
    int8_t *__restrict__ cw = conf_slice + (long)w * conf_stride;
    double *uit = UiT + (size_t)b * KD * KD, *vit = ViT + (size_t)b * KD * KD;

    const int sl = t - site0;  // my index inside the chunk, if any
    // FULL: wave-uniform (the wave whose first row is site0), decided on a scalar
    const bool in_chunk = FULL ? (__builtin_amdgcn_readfirstlane(t) == site0) : (active && sl >= 0 && sl < nsites);
    // the diagonal entry of my row (if it is one of the chunk's sites) is kept in a register: I am its only
    // writer, the LDS copy is what the proposals of all threads read
    double dgv = 0.0;
    if (in_chunk) { dgv = G[t + (long)n * t]; dg[b * KD + sl] = dgv; }
    const WalkerRng rs = rngs[w];
    if (tid < KD) flip[tid] = 0;
    if (tid < nsites) {
        cs[tid] = cw[site0 + tid];
        // the k-th uniform consumed by this chunk, whichever site consumes it (DQMC.jl:573)
        const unsigned long long d = rs.draw + (unsigned long long)tid;
        ul[tid] = rs.uniforms ? (d < rs.n_uniforms ? rs.uniforms[d] : 2.0) : philox_uniform(rs.seed, d);
    }
    int ndraw = 0, nneg = 0, exhausted = 0, cnt = 0;
    double u_next = 0.0;  // ul[ndraw], requested as soon as its predecessor has been consumed
    const double g0 = sc.gamma[0], g1 = sc.gamma[1], e0 = sc.ebos[0], e1 = sc.ebos[1];
    const double du0 = sc.dup[0], du1 = sc.dup[1], dd0 = sc.ddn[0], dd1 = sc.ddn[1];

    // G0[:, site] and G0[site, :] for one group of sites, requested one group ahead
    double colr[SW_GROUP], rowr[SW_GROUP], coln[SW_GROUP], rown[SW_GROUP];
    // no per-lane predicates: padding lanes read row/column n-1, sites past the chunk re-read the last one
    // (their values are never used); a group that lies entirely past the chunk is skipped by a uniform branch
    const int tq = FULL ? t : (active ? t : n - 1);
    auto fetch = [&](int s0, double (&cc)[SW_GROUP], double (&rr)[SW_GROUP]) {
        if (s0 >= nsites) return;
        if (FULL) {
            // the 8 row entries G[site0+s0 .. +7, t] are contiguous and 16-byte aligned: four wide loads instead of
            // eight narrow ones (each such instruction touches 64 different lines per wave)
            const double2 *rp = reinterpret_cast<const double2 *>(G + site0 + s0 + (long)n * tq);
#pragma unroll
            for (int x = 0; x < SW_GROUP / 2; ++x) {
                const double2 v = rp[x];
                rr[2 * x] = v.x;
                rr[2 * x + 1] = v.y;
            }
#pragma unroll
            for (int q = 0; q < SW_GROUP; ++q) cc[q] = G[tq + (long)n * (site0 + s0 + q)];
            return;
        }
#pragma unroll
        for (int q = 0; q < SW_GROUP; ++q) {
            const int site = site0 + min(s0 + q, nsites - 1);
            cc[q] = G[tq + (long)n * site];
            rr[q] = G[site + (long)n * tq];
        }
    };
    fetch(0, colr, rowr);
    __syncthreads();
    u_next = ul[0];

    for (int s0 = 0; s0 < nsites; s0 += SW_GROUP) {
        fetch(s0 + SW_GROUP, coln, rown);
        // ---- group start: bring the group's rows/columns up to date with all slots of the chunk
        {
            const int cnt0 = __builtin_amdgcn_readfirstlane(cnt);
            constexpr int HB = 8;  // slots per batch; the next batch is in flight while this one is consumed
            // padding lanes (t >= n) read row n-1 and slots past the end re-read slot cnt0-1: the loads carry
            // no predicates; only the arithmetic of the last, partial batch sits behind wave-uniform branches
            const double *__restrict__ hU = Uo + (active ? t : n - 1);
            const double *__restrict__ hV = VTo + (active ? t : n - 1);
            double hu[HB], hv[HB], hun[HB], hvn[HB];
            auto load_batch = [&](int m0, double (&au)[HB], double (&av)[HB]) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const int m = min(m0 + k, cnt0 - 1);
                    au[k] = hU[(long)n * m];
                    av[k] = hV[(long)n * m];
                }
            };
            if (cnt0 > 0) load_batch(0, hu, hv);
            int mb = 0;
            for (; mb + HB <= cnt0; mb += HB) {  // full batches: no per-slot tests
                if (mb + HB < cnt0) load_batch(mb + HB, hun, hvn);
                // LDS operands buffered three slots ahead: their reads are in flight during the current slot's FMAs
                double2 opu[4][SW_GROUP / 2], opv[4][SW_GROUP / 2];
                auto load_ops = [&](int m, double2 (&ou)[SW_GROUP / 2], double2 (&ov)[SW_GROUP / 2]) {
                    const double2 *ub = reinterpret_cast<const double2 *>(uit + m * KD + s0);
                    const double2 *vb = reinterpret_cast<const double2 *>(vit + m * KD + s0);
#pragma unroll
                    for (int x = 0; x < SW_GROUP / 2; ++x) { ou[x] = ub[x]; ov[x] = vb[x]; }
                };
                load_ops(mb, opu[0], opv[0]);
                load_ops(mb + 1, opu[1], opv[1]);
                load_ops(mb + 2, opu[2], opv[2]);
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    if (k + 3 < HB) load_ops(mb + k + 3, opu[(k + 3) & 3], opv[(k + 3) & 3]);
#pragma unroll
                    for (int x = 0; x < SW_GROUP / 2; ++x) {
                        colr[2 * x] += hu[k] * opv[k & 3][x].x;       // G[t, site_q] += U'[t][m] V[m][site_q]
                        colr[2 * x + 1] += hu[k] * opv[k & 3][x].y;
                        rowr[2 * x] += opu[k & 3][x].x * hv[k];       // G[site_q, t] += U'[site_q][m] V[m][t]
                        rowr[2 * x + 1] += opu[k & 3][x].y * hv[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < HB; ++k) { hu[k] = hun[k]; hv[k] = hvn[k]; }
            }
            if (mb < cnt0) {  // last, partial batch (already loaded; slots past the end are clamped copies)
#pragma unroll
                for (int k = 0; k < HB - 1; ++k) {
                    if (mb + k < cnt0) {
                        const double *ub = uit + (mb + k) * KD + s0, *vb = vit + (mb + k) * KD + s0;
#pragma unroll
                        for (int q = 0; q < SW_GROUP; ++q) {
                            colr[q] += hu[k] * vb[q];
                            rowr[q] += ub[q] * hv[k];
                        }
                    }
                }
            }
        }
        // the next site's spin and diagonal entries are requested one site ahead: after the barrier of an
        // accept (they are final then: an accept only touches entries of later sites) or right away on a reject
        int c_in = cs[s0];
        double d0_in = dg[s0], d1_in = MODEL != 0 ? dg[KD + s0] : 0.0;
#pragma unroll
        for (int q = 0; q < SW_GROUP; ++q) {
            const int s = s0 + q;
            if (FULL || s < nsites) {
                const int i = site0 + s;
                const int c = c_in;
                const int ci = c > 0 ? 1 : 0;
                const double d0 = d0_in;
                double detratio, p, x0, x1 = 0.0, r0s = 0.0, r1s = 0.0, d0s = 0.0, d1s = 0.0;
                if (model == 0) {  // HubbardModelAttractive.jl:113-127
                    const double gamma = ci ? g1 : g0;
                    const double r = 1.0 + gamma * (1.0 - d0);
                    detratio = r * r;
                    p = (ci ? e1 : e0) * detratio;
                    x0 = gamma;  // numerator; the division by r is off the decision path (done on accept)
                    x1 = r;
                } else {           // HubbardModelRepulsive.jl:128-156,174-191
                    const double d1 = d1_in;
                    const double D0 = ci ? du1 : du0, D1 = ci ? dd1 : dd0;
                    const double R0 = 1.0 + D0 * (1.0 - d0), R1 = 1.0 + D1 * (1.0 - d1);
                    detratio = R0 * R1;
                    p = detratio;
                    r0s = R0; r1s = R1; d0s = D0; d1s = D1;
                }
                if (MODEL != 0 && check_sign && detratio < 0.0) {  // attractive: detratio = r^2 >= 0
                    if (tid == 0) negv[nneg] = detratio;
                    ++nneg;
                }
                bool acc;
                if (p > 1.0) acc = true;  // DQMC.jl:573: rand() is consumed only when p <= 1
                else {
                    const double u = u_next;
                    ++ndraw;
                    u_next = ul[ndraw];  // (one past the chunk's last uniform at most: inside the LDS block, never used)
                    if (u == 2.0) exhausted = 1;
                    acc = u < p;
                }
                if (acc) {
                    const int j = __builtin_amdgcn_readfirstlane(cnt);
                    double xb;
                    if (model == 0) xb = x0 / x1;  // Attractive.jl:149: x = gamma / (1 + gamma*IG[i])
                    else {                          // Repulsive.jl:174-191: (R_other * inv_div) * Delta_b
                        const double inv_div = 1.0 / detratio;
                        xb = (b == 0) ? (r1s * inv_div) * d0s : (r0s * inv_div) * d1s;
                    }
                    const double ut = ((t == i) ? 1.0 : 0.0) - colr[q];  // IG = e_i - G[:,i]
                    double newU = -(ut * xb), newV = rowr[q];
                    if (!active) { newU = 0.0; newV = 0.0; }
                    if (active) {  // slot history of this thread (also the flush GEMM's operands)
                        Uo[t + (long)n * j] = newU;
                        VTo[t + (long)n * j] = newV;
                    }
                    if (in_chunk) {
                        uit[j * KD + sl] = newU;
                        vit[j * KD + sl] = newV;
                        if (sl > s) { dgv += newU * newV; dg[b * KD + sl] = dgv; }  // sites <= s are done
                    }
                    // cs[s] itself must stay intact: waves drift apart between barriers (a rejected
                    // site has none) and a slower wave may not have read it yet for ITS proposal
                    if (FULL) { if (__builtin_amdgcn_readfirstlane(tid) == 0) flip[s] = 1; }  // wave 0, same value from every lane
                    else if (tid == 0) flip[s] = 1;
                    ++cnt;
                    // LDS-only barrier: __syncthreads() would drain the history stores as well
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (q + 1 < SW_GROUP) {  // (reads one entry past the chunk at most: inside the LDS block, unused)
                        c_in = cs[s + 1];
                        d0_in = dg[s + 1];
                        if (MODEL != 0) d1_in = dg[KD + s + 1];
                    }
                    // eager update of the rest of the group (static q' > q)
                    {
                        const double *ub = uit + j * KD + s0, *vb = vit + j * KD + s0;
#pragma unroll
                        for (int q2 = q + 1; q2 < SW_GROUP; ++q2) {
                            colr[q2] += newU * vb[q2];
                            rowr[q2] += ub[q2] * newV;
                        }
                    }
                } else if (q + 1 < SW_GROUP) {
                    c_in = cs[s + 1];
                    d0_in = dg[s + 1];
                    if (MODEL != 0) d1_in = dg[KD + s + 1];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < SW_GROUP; ++q) { colr[q] = coln[q]; rowr[q] = rown[q]; }
    }
    // unused slots are zero padded so that the flush GEMM G0 += U' V always runs with K = KD
    if (active) {
        for (int m = cnt; m < KD; ++m) {
            Uo[t + (long)n * m] = 0.0;
            VTo[t + (long)n * m] = 0.0;
        }
    }
    if (tid < nsites) cw[site0 + tid] = (int8_t)(flip[tid] ? -cs[tid] : cs[tid]);
    if (tid == 0) {
        for (int k = 0; k < nneg; ++k) magstats_push(stats[w].negative_probability, negv[k]);
        rngs[w].draw = rs.draw + (unsigned long long)ndraw;
        if (exhausted) rngs[w].exhausted = 1;
        stats[w].prop_local += nsites;
        stats[w].acc_local += cnt;
    }
}

// chunk length = update slots per flush (the history lives in global memory, LDS holds KD x KD tables)
int sweep_kd(int n, int nb) { (void)n; (void)nb; return SW_KD; }

hipError_t launch_sweep_chunk(int n, int nb, int n_walkers, int model, double *G, long strideG, int8_t *conf_slice,
                              long conf_stride, int site0, int nsites, double *Uout, double *VTout, long strideUV,
                              SweepConsts sc, WalkerRng *rng, DevStats *stats, int check_sign, hipStream_t s)
{
    const int npad = (n + 63) & ~63;
    const int threads = nb * npad;
    if (threads > 1024 || nsites > SW_KD) return hipErrorInvalidValue;
    dim3 grid(n_walkers), block(threads);
    const size_t lds = ((size_t)nb * 2 * SW_KD * SW_KD + 6 * SW_KD) * sizeof(double) + 64;
    const bool full = (n % 64 == 0) && (site0 % 64 == 0) && nsites == 64;
#define SW_LAUNCH3(MT, MD, FL)                                                                                   \
    do {                                                                                                         \
        (void)hipFuncSetAttribute((const void *)sweep_chunk_kernel<MT, MD, FL>,                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); /* per device: every launch */ \
        hipLaunchKernelGGL((sweep_chunk_kernel<MT, MD, FL>), grid, block, lds, s, n, nb, model, G, strideG,      \
                           conf_slice, conf_stride, site0, nsites, Uout, VTout, strideUV, sc, rng, stats,        \
                           check_sign);                                                                          \
    } while (0)
#define SW_LAUNCH(MT)                                                                                            \
    do {                                                                                                         \
        if (model == 0) { if (full) SW_LAUNCH3(MT, 0, true); else SW_LAUNCH3(MT, 0, false); }                    \
        else { if (full) SW_LAUNCH3(MT, 1, true); else SW_LAUNCH3(MT, 1, false); }                               \
    } while (0)
    if (threads <= 256) SW_LAUNCH(256);
    else if (threads <= 512) SW_LAUNCH(512);
    else SW_LAUNCH(1024);
#undef SW_LAUNCH
#undef SW_LAUNCH3
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ void set_identity_kernel(int n, double *A, long stride)
{
    double *a = A + (long)blockIdx.y * stride;
    const long nn = (long)n * n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nn; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        a[idx] = r == c ? 1.0 : 0.0;
    }
}
hipError_t launch_set_identity(int n, int count, double *A, long stride, hipStream_t s)
{
    const long nn = (long)n * n;
    int bx = (int)((nn + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(set_identity_kernel, dim3(bx, count), dim3(256), 0, s, n, A, stride);
    return hipGetLastError();
}
__global__ void set_diag_kernel(int n, int nb, double *A, long stride, VecSrc d)
{
    const int unit = blockIdx.y;
    double *a = A + (long)unit * stride;
    const long nn = (long)n * n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nn; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        a[idx] = r == c ? vs_get(d, unit, nb, r) : 0.0;
    }
}
hipError_t launch_set_diag(int n, int nb, int units, double *A, long stride, VecSrc d, hipStream_t s)
{
    const long nn = (long)n * n;
    int bx = (int)((nn + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(set_diag_kernel, dim3(bx, units), dim3(256), 0, s, n, nb, A, stride, d);
    return hipGetLastError();
}
__global__ void mat_add_kernel(double *__restrict__ A, const double *__restrict__ B, size_t count)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        A[i] += B[i];
}
hipError_t launch_mat_add(double *A, const double *B, size_t count, hipStream_t s)
{
    size_t b = (count + 255) / 256;
    if (b > 4096) b = 4096;
    if (b == 0) b = 1;
    hipLaunchKernelGGL(mat_add_kernel, dim3((unsigned)b), dim3(256), 0, s, A, B, count);
    return hipGetLastError();
}
__global__ void sub_identity_kernel(int n, double *__restrict__ O, const double *__restrict__ A, long stride)
{
    double *o = O + (long)blockIdx.y * stride;
    const double *a = A + (long)blockIdx.y * stride;
    const long nn = (long)n * n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nn; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        o[idx] = a[idx] - (r == c ? 1.0 : 0.0);
    }
}
hipError_t launch_sub_identity(int n, int units, double *O, const double *A, long stride, hipStream_t s)
{
    const long nn = (long)n * n;
    int bx = (int)((nn + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(sub_identity_kernel, dim3(bx, units), dim3(256), 0, s, n, O, A, stride);
    return hipGetLastError();
}
// O = Diagonal(row) * A * Diagonal(col), the two scalings applied in the reference's order
__global__ void scale_mat_kernel(int n, int nb, double *__restrict__ O, const double *__restrict__ A, long stride,
                                 VecSrc row, VecSrc col, int row_first)
{
    const int unit = blockIdx.y;
    double *o = O + (long)unit * stride;
    const double *a = A + (long)unit * stride;
    const long nn = (long)n * n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nn; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        const double rs = row.mode ? vs_get(row, unit, nb, r) : 1.0, cs = col.mode ? vs_get(col, unit, nb, c) : 1.0;
        double v = a[idx];
        if (row_first) { v *= rs; v *= cs; } else { v *= cs; v *= rs; }
        o[idx] = v;
    }
}
hipError_t launch_scale_mat(int n, int nb, int units, double *O, const double *A, long stride, VecSrc row, VecSrc col,
                            int row_first, hipStream_t s)
{
    const long nn = (long)n * n;
    int bx = (int)((nn + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(scale_mat_kernel, dim3(bx, units), dim3(256), 0, s, n, nb, O, A, stride, row, col, row_first);
    return hipGetLastError();
}
__global__ void vec_map_kernel(int n, int nb, double *dst, long stride, VecSrc src)
{
    const int unit = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = vs_get(src, unit, nb, i);  // src may alias dst: each element is read, then written
        dst[(long)unit * stride + i] = v;
    }
}
hipError_t launch_vec_map(int n, int nb, int units, double *dst, long stride, VecSrc src, hipStream_t s)
{
    hipLaunchKernelGGL(vec_map_kernel, dim3(units), dim3(256), 0, s, n, nb, dst, stride, src);
    return hipGetLastError();
}
__global__ void fill_kernel(double *p, size_t n, double v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t launch_fill(double *p, size_t n, double v, hipStream_t s)
{
    size_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b == 0) b = 1;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)b), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

// maximum(abs.(greens_temp - greens)) > 1e-7 -> push!(propagation_error, .) (stack.jl:538-549,602-611)
// PC_SPLIT workgroups per walker: partial maxima meet in a per-walker word (non-negative doubles order like their bit
// patterns), the last workgroup to arrive pushes the event and clears the scratch for the next launch
constexpr int PC_SPLIT = 8;
__global__ __launch_bounds__(1024) void prop_check_kernel(int n, int nb, const double *__restrict__ A,
                                                         const double *__restrict__ B, long stride_unit,
                                                         DevStats *stats, unsigned long long *scratch)
{
    __shared__ double red[16];
    const int w = blockIdx.x / PC_SPLIT, part = blockIdx.x % PC_SPLIT;
    const long tot = (long)nb * stride_unit;
    double d = 0.0;
    bool bad = false;  // a NaN anywhere makes the maximum NaN in the reference (maximum(abs.(...)))
    if (tot & 1) {     // odd element count (odd n): the walkers' matrices are not all 16-byte aligned
        const double *a = A + (long)w * tot, *b = B + (long)w * tot;
        for (long i = (long)part * blockDim.x + threadIdx.x; i < tot; i += (long)PC_SPLIT * blockDim.x) {
            const double e0 = fabs(a[i] - b[i]);
            bad |= (e0 != e0);
            d = fmax(d, e0);
        }
    } else {
        const double2 *a = reinterpret_cast<const double2 *>(A + (long)w * tot), *b = reinterpret_cast<const double2 *>(B + (long)w * tot);
        for (long i = (long)part * blockDim.x + threadIdx.x; i < tot / 2; i += (long)PC_SPLIT * blockDim.x) {
            const double2 x = a[i], y = b[i];
            const double e0 = fabs(x.x - y.x), e1 = fabs(x.y - y.y);
            bad |= (e0 != e0) | (e1 != e1);
            d = fmax(d, fmax(e0, e1));
        }
    }
    if (bad) d = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(d, off, 64);
        d = (d != d || o != o) ? __longlong_as_double(0x7ff8000000000000ll) : fmax(d, o);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) d = (d != d || red[i] != red[i]) ? __longlong_as_double(0x7ff8000000000000ll) : fmax(d, red[i]);
        // NaN (0x7ff8...) sorts above every finite magnitude as an unsigned word: it survives the atomicMax
        atomicMax(&scratch[2 * w], (unsigned long long)__double_as_longlong(d));
        __threadfence();
        if (atomicAdd(&scratch[2 * w + 1], 1ull) == PC_SPLIT - 1) {
            __threadfence();
            const double m = __longlong_as_double((long long)atomicExch(&scratch[2 * w], 0ull));
            scratch[2 * w + 1] = 0ull;
            if (m > 1e-7) magstats_push(stats[w].propagation_error, m);
        }
    }
}
hipError_t launch_prop_check(int n, int nb, int n_walkers, const double *A, const double *B, long stride_unit,
                             DevStats *stats, unsigned long long *scratch, hipStream_t s)
{
    hipLaunchKernelGGL(prop_check_kernel, dim3(n_walkers * PC_SPLIT), dim3(1024), 0, s, n, nb, A, B, stride_unit, stats,
                       scratch);
    return hipGetLastError();
}

// Measurement sums over the walkers of this device, fixed walker order (deterministic).
__global__ void accumulate_kernel(int n, int nb, int n_walkers, const double *__restrict__ G, long stride_unit,
                                  double *__restrict__ acc)
{
    const long per = (long)nb * n * n;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / ((long)n * n));
        const long idx = e - (long)b * n * n;
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < n_walkers; ++w) {
            const double g = G[((long)w * nb + b) * stride_unit + idx];
            s1 += g;
            s2 += g * g;
        }
        acc[e] += s1;
        acc[per + e] += s2;
        const int r = (int)(idx % n), c = (int)(idx / n);
        if (r == c) acc[2 * per + (long)b * n + r] += (double)n_walkers - s1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[2 * per + (long)nb * n] += (double)n_walkers;
}
hipError_t launch_accumulate(int n, int nb, int n_walkers, const double *G, long stride_unit, double *acc,
                             hipStream_t s)
{
    const long per = (long)nb * n * n;
    int bx = (int)((per + 255) / 256);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(accumulate_kernel, dim3(bx), dim3(256), 0, s, n, nb, n_walkers, G, stride_unit, acc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Equal-time correlation measurements on the true Green's function:
// cdc_kernel, sdc_{x,y,z}_kernel over EachSitePairByDistance and m{x,y,z}_kernel over EachSite
// (src/flavors/DQMC/measurements/measurements.jl:51-190, generic.jl:325-330,
// attractive overrides HubbardModelAttractive.jl:219-246).  For the repulsive model G is block
// diagonal (up, down), so every cross-spin element of the reference's 2N x 2N formulas is 0.
// One workgroup sums one direction of one walker over its pairs in a fixed order (deterministic).
__global__ __launch_bounds__(256) void corr_pairs_kernel(int n, int nb, int model, const double *__restrict__ G,
                                                        long stride_unit, const int *__restrict__ dir_ptr,
                                                        const int *__restrict__ pair_src,
                                                        const int *__restrict__ pair_trg, int n_dirs,
                                                        double *__restrict__ per_walker, long per_stride)
{
    __shared__ double red[4][256];
    const int d = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const double *G1 = G + (long)(w * nb) * stride_unit;
    const double *G2 = nb == 2 ? G1 + stride_unit : G1;
    double cdc = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (int q = dir_ptr[d] + tid; q < dir_ptr[d + 1]; q += 256) {
        const int i = pair_src[q], j = pair_trg[q];
        const double dij = i == j ? 1.0 : 0.0;
        const double a_ii = G1[i + (long)n * i], a_jj = G1[j + (long)n * j];
        const double a_ij = G1[i + (long)n * j], a_ji = G1[j + (long)n * i];
        if (model == 0) {  // HubbardModelAttractive.jl:222-236
            const double t = 2.0 * (dij - a_ji) * a_ij;
            cdc += 4.0 * (1.0 - a_ii) * (1.0 - a_jj) + t;
            sx += t; sy += t; sz += t;
        } else {
            const double b_ii = G2[i + (long)n * i], b_jj = G2[j + (long)n * j];
            const double b_ij = G2[i + (long)n * j], b_ji = G2[j + (long)n * i];
            // measurements.jl:63-76
            cdc += (1.0 - a_ii) * (1.0 - a_jj) + (dij - a_ji) * a_ij + (1.0 - a_ii) * (1.0 - b_jj) +
                   (1.0 - b_ii) * (1.0 - a_jj) + (1.0 - b_ii) * (1.0 - b_jj) + (dij - b_ji) * b_ij;
            // measurements.jl:150-156 / :167-173 with the cross-spin elements zero
            const double t = (dij - a_ji) * b_ij + (dij - b_ji) * a_ij;
            sx += t; sy += t;
            // measurements.jl:184-189
            sz += (1.0 - a_ii) * (1.0 - a_jj) + (dij - a_ji) * a_ij - (1.0 - a_ii) * (1.0 - b_jj) -
                  (1.0 - b_ii) * (1.0 - a_jj) + (1.0 - b_ii) * (1.0 - b_jj) + (dij - b_ji) * b_ij;
        }
    }
    red[0][tid] = cdc; red[1][tid] = sx; red[2][tid] = sy; red[3][tid] = sz;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid < 4) per_walker[(long)w * per_stride + (long)tid * n_dirs + d] = red[tid][0] / (double)n;  // finish!: / N
}
// m{x,y,z}_kernel (measurements.jl:112-124; attractive: all zero) and the sum over walkers
__global__ void corr_reduce_kernel(int n, int nb, int model, int n_walkers, const double *__restrict__ G,
                                   long stride_unit, int n_dirs, const double *__restrict__ per_walker,
                                   long per_stride, double *__restrict__ acc)
{
    const int total = 4 * n_dirs + 3 * n;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        double s = 0.0;
        if (e < 4 * n_dirs) {
            for (int w = 0; w < n_walkers; ++w) s += per_walker[(long)w * per_stride + e];
        } else if (e >= 4 * n_dirs + 2 * n && model != 0) {  // mz = G_dn[i,i] - G_up[i,i]
            const int i = e - 4 * n_dirs - 2 * n;
            for (int w = 0; w < n_walkers; ++w) {
                const double *G1 = G + (long)(w * nb) * stride_unit;
                s += G1[stride_unit + i + (long)n * i] - G1[i + (long)n * i];
            }
        }
        acc[e] += s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[total] += (double)n_walkers;
}
hipError_t launch_correlations(int n, int nb, int model, int n_walkers, const double *G, long stride_unit,
                               const int *dir_ptr, const int *pair_src, const int *pair_trg, int n_dirs,
                               double *per_walker, double *acc, hipStream_t s)
{
    const long per_stride = 4L * n_dirs;
    hipLaunchKernelGGL(corr_pairs_kernel, dim3(n_dirs, n_walkers), dim3(256), 0, s, n, nb, model, G, stride_unit,
                       dir_ptr, pair_src, pair_trg, n_dirs, per_walker, per_stride);
    const int total = 4 * n_dirs + 3 * n;
    hipLaunchKernelGGL(corr_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, n, nb, model, n_walkers, G,
                       stride_unit, n_dirs, per_walker, per_stride, acc);
    return hipGetLastError();
}

// pc_kernel over EachLocalQuadByDistance{K} (measurements.jl:208-214, generic.jl:341-349,
// lattice_iterators.jl:264-318; attractive override HubbardModelAttractive.jl:243-245):
//   out[dir12, dir1, dir2] += G[src1, src2] * G[trg1+N, trg2+N] - G[src1, trg2+N] * G[trg1+N, src2]
// summed over all (src1, src2) with direction dir12 and the targets trg_k(src) of src in the K
// shortest directions.  With block-diagonal G (both Hubbard models here) the second product is 0 and
// the first is G_up[src1,src2] * G_dn[trg1,trg2] (attractive: G_dn = G_up).  One workgroup owns one
// (direction, walker) and reduces every (dir1, dir2) in a fixed order (deterministic).
__global__ __launch_bounds__(256) void pairing_kernel(int n, int nb, const double *__restrict__ G, long stride_unit,
                                                     const int *__restrict__ dir_ptr,
                                                     const int *__restrict__ pair_src,
                                                     const int *__restrict__ pair_trg, int n_dirs, int K,
                                                     const int *__restrict__ trg_of,
                                                     double *__restrict__ per_walker, long per_stride)
{
    __shared__ double red[256];
    const int d = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const double *G1 = G + (long)(w * nb) * stride_unit;
    const double *G2 = nb == 2 ? G1 + stride_unit : G1;
    for (int k2 = 0; k2 < K; ++k2)
        for (int k1 = 0; k1 < K; ++k1) {
            double s = 0.0;
            for (int q = dir_ptr[d] + tid; q < dir_ptr[d + 1]; q += 256) {
                const int s1 = pair_src[q], s2 = pair_trg[q];
                const int t1 = trg_of[s1 + n * k1], t2 = trg_of[s2 + n * k2];
                if (t1 >= 0 && t2 >= 0) s += G1[s1 + (long)n * s2] * G2[t1 + (long)n * t2];
            }
            red[tid] = s;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (tid < off) red[tid] += red[tid + off];
                __syncthreads();
            }
            if (tid == 0)  // finish!: / N (generic.jl:287-290); Julia layout [dir12, dir1, dir2]
                per_walker[(long)w * per_stride + d + (long)n_dirs * (k1 + K * k2)] = red[0] / (double)n;
            __syncthreads();
        }
}
__global__ void pairing_reduce_kernel(int n_walkers, long total, const double *__restrict__ per_walker,
                                      double *__restrict__ acc)
{
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int w = 0; w < n_walkers; ++w) s += per_walker[(long)w * total + e];
        acc[e] += s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[total] += (double)n_walkers;
}
hipError_t launch_pairing(int n, int nb, int n_walkers, const double *G, long stride_unit, const int *dir_ptr,
                          const int *pair_src, const int *pair_trg, int n_dirs, int K, const int *trg_of,
                          double *per_walker, double *acc, hipStream_t s)
{
    const long total = (long)n_dirs * K * K;
    hipLaunchKernelGGL(pairing_kernel, dim3(n_dirs, n_walkers), dim3(256), 0, s, n, nb, G, stride_unit, dir_ptr,
                       pair_src, pair_trg, n_dirs, K, trg_of, per_walker, total);
    hipLaunchKernelGGL(pairing_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n_walkers, total,
                       per_walker, acc);
    return hipGetLastError();
}

// Time-displaced versions (packed Green's functions (G00, G0l, Gl0, Gll), measurements.jl:76-92,
// 150-192, 215-219; attractive overrides HubbardModelAttractive.jl:226-246) for the susceptibilities
// that apply!(::CombinedGreensIterator, ...) integrates over l (generic.jl:226-243).  Block-diagonal G:
// every cross-spin element of the 2N x 2N formulas is 0.  Each launch ADDS the contribution of one
// time slice to the per-walker sums (fixed order inside a workgroup: deterministic).
__global__ __launch_bounds__(256) void sus_pairs_kernel(int n, int nb, int model, const double *__restrict__ G00,
                                                       const double *__restrict__ G0l,
                                                       const double *__restrict__ Gl0,
                                                       const double *__restrict__ Gll, long stride_unit,
                                                       const int *__restrict__ dir_ptr,
                                                       const int *__restrict__ pair_src,
                                                       const int *__restrict__ pair_trg, int n_dirs,
                                                       double *__restrict__ per_walker, long per_stride)
{
    __shared__ double red[3][256];
    const int d = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const long u0 = (long)(w * nb) * stride_unit, u1 = nb == 2 ? u0 + stride_unit : u0;
    double cdc = 0.0, sxy = 0.0, sz = 0.0;
    for (int q = dir_ptr[d] + tid; q < dir_ptr[d + 1]; q += 256) {
        const int i = pair_src[q], j = pair_trg[q];
        const double l_up = 1.0 - Gll[u0 + i + (long)n * i], z_up = 1.0 - G00[u0 + j + (long)n * j];
        const double x_up = G0l[u0 + j + (long)n * i] * Gl0[u0 + i + (long)n * j];
        if (model == 0) {
            cdc += 4.0 * l_up * z_up - 2.0 * x_up;
            sxy += -2.0 * x_up;
            sz += -2.0 * x_up;
        } else {
            const double l_dn = 1.0 - Gll[u1 + i + (long)n * i], z_dn = 1.0 - G00[u1 + j + (long)n * j];
            const double x_dn = G0l[u1 + j + (long)n * i] * Gl0[u1 + i + (long)n * j];
            cdc += l_up * z_up - x_up + l_up * z_dn + l_dn * z_up + l_dn * z_dn - x_dn;
            // - G0l[j, i] Gl0[i+N, j+N] - G0l[j+N, i+N] Gl0[i, j]
            sxy += -G0l[u0 + j + (long)n * i] * Gl0[u1 + i + (long)n * j] - G0l[u1 + j + (long)n * i] * Gl0[u0 + i + (long)n * j];
            sz += l_up * z_up - x_up - l_up * z_dn - l_dn * z_up + l_dn * z_dn - x_dn;
        }
    }
    red[0][tid] = cdc; red[1][tid] = sxy; red[2][tid] = sz;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
            for (int q = 0; q < 3; ++q) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid < 4) {  // [cds][sds_x][sds_y][sds_z]; x and y coincide for block-diagonal G
        const int src = tid == 0 ? 0 : (tid == 3 ? 2 : 1);
        per_walker[(long)w * per_stride + (long)tid * n_dirs + d] += red[src][0] / (double)n;
    }
}
// pc_kernel with packed Green's functions: Gl0[src1, src2] * Gl0[trg1+N, trg2+N] (- cross-spin term = 0)
__global__ __launch_bounds__(256) void sus_pairing_kernel(int n, int nb, const double *__restrict__ Gl0,
                                                         long stride_unit, const int *__restrict__ dir_ptr,
                                                         const int *__restrict__ pair_src,
                                                         const int *__restrict__ pair_trg, int n_dirs, int K,
                                                         const int *__restrict__ trg_of,
                                                         double *__restrict__ per_walker, long per_stride, long offset)
{
    __shared__ double red[256];
    const int d = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const double *G1 = Gl0 + (long)(w * nb) * stride_unit;
    const double *G2 = nb == 2 ? G1 + stride_unit : G1;
    for (int k2 = 0; k2 < K; ++k2)
        for (int k1 = 0; k1 < K; ++k1) {
            double s = 0.0;
            for (int q = dir_ptr[d] + tid; q < dir_ptr[d + 1]; q += 256) {
                const int s1 = pair_src[q], s2 = pair_trg[q];
                const int t1 = trg_of[s1 + n * k1], t2 = trg_of[s2 + n * k2];
                if (t1 >= 0 && t2 >= 0) s += G1[s1 + (long)n * s2] * G2[t1 + (long)n * t2];
            }
            red[tid] = s;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (tid < off) red[tid] += red[tid + off];
                __syncthreads();
            }
            if (tid == 0) per_walker[(long)w * per_stride + offset + d + (long)n_dirs * (k1 + K * k2)] += red[0] / (double)n;
            __syncthreads();
        }
}
// acc[e] += factor * sum_w per_walker[w][e]; acc[total] += n_walkers
__global__ void sus_reduce_kernel(int n_walkers, long total, double factor, const double *__restrict__ per_walker,
                                  double *__restrict__ acc)
{
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int w = 0; w < n_walkers; ++w) s += per_walker[(long)w * total + e];
        acc[e] += factor * s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[total] += (double)n_walkers;
}
hipError_t launch_sus_slice(int n, int nb, int model, int n_walkers, const double *G00, const double *G0l,
                            const double *Gl0, const double *Gll, long stride_unit, const int *dir_ptr,
                            const int *pair_src, const int *pair_trg, int n_dirs, int K, const int *trg_of,
                            double *per_walker, long per_stride, hipStream_t s)
{
    hipLaunchKernelGGL(sus_pairs_kernel, dim3(n_dirs, n_walkers), dim3(256), 0, s, n, nb, model, G00, G0l, Gl0, Gll,
                       stride_unit, dir_ptr, pair_src, pair_trg, n_dirs, per_walker, per_stride);
    if (K > 0)
        hipLaunchKernelGGL(sus_pairing_kernel, dim3(n_dirs, n_walkers), dim3(256), 0, s, n, nb, Gl0, stride_unit,
                           dir_ptr, pair_src, pair_trg, n_dirs, K, trg_of, per_walker, per_stride, 4L * n_dirs);
    return hipGetLastError();
}
hipError_t launch_sus_reduce(int n_walkers, long total, double factor, const double *per_walker, double *acc,
                             hipStream_t s)
{
    hipLaunchKernelGGL(sus_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n_walkers, total,
                       factor, per_walker, acc);
    return hipGetLastError();
}

// compress(mc, model, conf) = BitArray(conf .== 1) (HubbardModel.jl:56-59): Julia's BitArray keeps
// element i (1-based, column-major) in bit (i-1) % 64 of chunk (i-1) / 64.  One wave packs one chunk.
__global__ void conf_pack_kernel(const int8_t *__restrict__ conf, size_t n_elem, unsigned long long *__restrict__ chunks)
{
    const size_t chunk = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t nchunks = (n_elem + 63) >> 6;
    if (chunk >= nchunks) return;
    const size_t i = chunk * 64 + (threadIdx.x & 63);
    const unsigned long long b = __ballot(i < n_elem && conf[i] == 1);
    if ((threadIdx.x & 63) == 0) chunks[chunk] = b;
}
// decompress: CT(2c .- 1)
__global__ void conf_unpack_kernel(const unsigned long long *__restrict__ chunks, size_t n_elem, int8_t *__restrict__ conf)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n_elem) conf[i] = ((chunks[i >> 6] >> (i & 63)) & 1ull) ? 1 : -1;
}
hipError_t launch_conf_pack(const int8_t *conf, size_t n_elem, unsigned long long *chunks, hipStream_t s)
{
    const size_t nchunks = (n_elem + 63) >> 6;
    hipLaunchKernelGGL(conf_pack_kernel, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, s, conf, n_elem, chunks);
    return hipGetLastError();
}
hipError_t launch_conf_unpack(const unsigned long long *chunks, size_t n_elem, int8_t *conf, hipStream_t s)
{
    hipLaunchKernelGGL(conf_unpack_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, s, chunks, n_elem, conf);
    return hipGetLastError();
}

}  // namespace dqmc
