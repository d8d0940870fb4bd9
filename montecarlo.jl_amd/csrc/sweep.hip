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

namespace dqmc {

__device__ __forceinline__ double philox_uniform(unsigned long long seed, unsigned long long index)
{
    unsigned int c0 = (unsigned int)index, c1 = (unsigned int)(index >> 32), c2 = 0u, c3 = 0u;
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned long long hi = c0 >> 5, lo = c1 >> 6;
    return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void magstats_push(DevMagStats &s, double value)
{
    const double v = log10(fabs(value));
    s.max = fmax(s.max, v);
    s.min = fmin(s.min, v);
    s.sum += v;
    s.count += 1;
}

template <int KD, int MAXT>
__global__ __launch_bounds__(MAXT) void sweep_chunk_kernel(int n, int nb, int model, double *__restrict__ Gall,
                                                          long strideG, int8_t *__restrict__ conf_slice,
                                                          long conf_stride, int site0, int nsites,
                                                          double *__restrict__ Uall, double *__restrict__ VTall,
                                                          long strideUV, SweepConsts sc, WalkerRng *rngs,
                                                          DevStats *stats, int check_sign)
{
    __shared__ double dg[2][KD];          // current G[i,i] of the chunk's sites
    __shared__ double Ui[2][KD][KD + 1];  // Ui[b][s][m] = U'[site0+s][m]
    __shared__ double Vi[2][KD][KD + 1];  // Vi[b][m][s] = V[m][site0+s]
    __shared__ int cs[KD];                // HS field of the chunk's sites

    const int w = blockIdx.x;
    const int npad = (n + 63) & ~63;
    const int tid = threadIdx.x;
    const int b = tid / npad, t = tid - b * npad;  // wave-uniform block index
    const bool active = t < n;
    const int unit = w * nb + b;
    double *__restrict__ G = Gall + (long)unit * strideG;
    double *__restrict__ Uo = Uall + (long)unit * strideUV;
    double *__restrict__ VTo = VTall + (long)unit * strideUV;
    int8_t *__restrict__ cw = conf_slice + (long)w * conf_stride;

    const int sl = t - site0;  // my index inside the chunk, if any
    const bool in_chunk = active && sl >= 0 && sl < nsites;
    if (in_chunk) dg[b][sl] = G[t + (long)n * t];
    if (tid < nsites) cs[tid] = cw[site0 + tid];
    WalkerRng rs = rngs[w];
    unsigned long long draw = rs.draw;
    int exhausted = 0;
    long long n_acc = 0;
    __syncthreads();

    double Ureg[KD], Vreg[KD];
#pragma unroll
    for (int m = 0; m < KD; ++m) { Ureg[m] = 0.0; Vreg[m] = 0.0; }
    int cnt = 0;

    double colc = 0.0, rowc = 0.0;
    if (active) {
        colc = G[t + (long)n * site0];
        rowc = G[site0 + (long)n * t];
    }
    for (int s = 0; s < nsites; ++s) {
        const int i = site0 + s;
        double coln = 0.0, rown = 0.0;
        if (active && s + 1 < nsites) {  // software prefetch of the next site's column / row of G0
            coln = G[t + (long)n * (i + 1)];
            rown = G[(i + 1) + (long)n * t];
        }
        const int c = cs[s];
        const int ci = c > 0 ? 1 : 0;
        const double d0 = dg[0][s];
        double detratio, p, x0, x1 = 0.0;
        if (model == 0) {  // HubbardModelAttractive.jl:113-127
            const double gamma = sc.gamma[ci];
            const double r = 1.0 + gamma * (1.0 - d0);
            detratio = r * r;
            p = sc.ebos[ci] * detratio;
            x0 = gamma / r;  // Attractive.jl:149: x = gamma / (1 + gamma*IG[i])
        } else {           // HubbardModelRepulsive.jl:128-156,174-191
            const double d1 = dg[1][s];
            const double D0 = sc.dup[ci], D1 = sc.ddn[ci];
            const double R0 = 1.0 + D0 * (1.0 - d0), R1 = 1.0 + D1 * (1.0 - d1);
            detratio = R0 * R1;
            p = detratio;
            const double inv_div = 1.0 / detratio;
            x0 = (R1 * inv_div) * D0;
            x1 = (R0 * inv_div) * D1;
        }
        if (check_sign && detratio < 0.0 && tid == 0) magstats_push(stats[w].negative_probability, detratio);

        bool acc;
        if (p > 1.0) acc = true;  // DQMC.jl:573: rand() is consumed only when p <= 1
        else {
            double u;
            if (rs.uniforms) {
                if (draw < rs.n_uniforms) u = rs.uniforms[draw];
                else { u = 2.0; exhausted = 1; }
            } else u = philox_uniform(rs.seed, draw);
            ++draw;
            acc = u < p;
        }
        if (acc) {
            const int j = __builtin_amdgcn_readfirstlane(cnt);
            const double xb = (b == 0) ? x0 : x1;
            double cold = colc, rowd = rowc;
#pragma unroll
            for (int m = 0; m < KD; ++m) {
                if (m < j) {
                    cold += Ureg[m] * Vi[b][m][s];
                    rowd += Ui[b][s][m] * Vreg[m];
                }
            }
            const double ut = ((t == i) ? 1.0 : 0.0) - cold;  // IG = e_i - G[:,i]
            double newU = -(ut * xb), newV = rowd;
            if (!active) { newU = 0.0; newV = 0.0; }
#pragma unroll
            for (int m = 0; m < KD; ++m) {
                if (m == j) { Ureg[m] = newU; Vreg[m] = newV; }
            }
            if (active) {
                Uo[t + (long)n * j] = newU;
                VTo[t + (long)n * j] = newV;
            }
            if (in_chunk && sl > s) {  // sites <= s are never proposed again in this chunk
                Ui[b][sl][j] = newU;
                Vi[b][j][sl] = newV;
                dg[b][sl] += newU * newV;
            }
            if (tid == 0) cw[i] = (int8_t)-c;
            ++cnt;
            ++n_acc;
            __syncthreads();
        }
        colc = coln;
        rowc = rown;
    }
    // zero the unused update slots so that the flush GEMM can always run with K = KD
    for (int m = cnt; m < KD; ++m) {
        if (active) {
            Uo[t + (long)n * m] = 0.0;
            VTo[t + (long)n * m] = 0.0;
        }
    }
    if (tid == 0) {
        rngs[w].draw = draw;
        if (exhausted) rngs[w].exhausted = 1;
        stats[w].prop_local += nsites;
        stats[w].acc_local += n_acc;
    }
}

// chunk length: 2*KD doubles of update vectors live in registers per thread, so the
// 1024-thread configuration (128 VGPR budget) uses the shorter chunk
int sweep_kd(int n, int nb) { return nb * ((n + 63) & ~63) <= 512 ? 32 : 16; }

hipError_t launch_sweep_chunk(int n, int nb, int n_walkers, int model, double *G, long strideG, int8_t *conf_slice,
                              long conf_stride, int site0, int nsites, double *Uout, double *VTout, long strideUV,
                              SweepConsts sc, WalkerRng *rng, DevStats *stats, int check_sign, hipStream_t s)
{
    const int npad = (n + 63) & ~63;
    const int threads = nb * npad;
    if (threads > 1024 || nsites > sweep_kd(n, nb)) return hipErrorInvalidValue;
    dim3 grid(n_walkers), block(threads);
#define SW_LAUNCH(KD, MT)                                                                                      \
    hipLaunchKernelGGL((sweep_chunk_kernel<KD, MT>), grid, block, 0, s, n, nb, model, G, strideG, conf_slice,  \
                       conf_stride, site0, nsites, Uout, VTout, strideUV, sc, rng, stats, check_sign)
    if (threads <= 256) SW_LAUNCH(32, 256);
    else if (threads <= 512) SW_LAUNCH(32, 512);
    else SW_LAUNCH(16, 1024);
#undef SW_LAUNCH
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
__global__ __launch_bounds__(1024) void prop_check_kernel(int n, int nb, const double *__restrict__ A,
                                                         const double *__restrict__ B, long stride_unit,
                                                         DevStats *stats)
{
    __shared__ double red[16];
    const int w = blockIdx.x;
    const long tot = (long)nb * stride_unit;
    const double *a = A + (long)w * tot, *b = B + (long)w * tot;
    double d = 0.0;
    for (long i = threadIdx.x; i < tot; i += blockDim.x) d = fmax(d, fabs(a[i] - b[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = fmax(d, __shfl_xor(d, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) d = fmax(d, red[i]);
        if (d > 1e-7) magstats_push(stats[w].propagation_error, d);
    }
}
hipError_t launch_prop_check(int n, int nb, int n_walkers, const double *A, const double *B, long stride_unit,
                             DevStats *stats, hipStream_t s)
{
    hipLaunchKernelGGL(prop_check_kernel, dim3(n_walkers), dim3(1024), 0, s, n, nb, A, B, stride_unit, stats);
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

}  // namespace dqmc
