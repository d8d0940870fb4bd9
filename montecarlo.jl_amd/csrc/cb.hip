// cb.hip — checkerboard (CheckerboardTrue) slice-matrix products with the bond-group factors kept SPARSE on the
// device: multiply_slice_matrix_left! / _right! / _inv_left! / _inv_right! / multiply_daggered_slice_matrix_left!
// (src/flavors/DQMC/slice_matrices.jl:104-222) and the half-step sandwiches of greens() (DQMC.jl:731-750).
//
// Every product is a short sequence of sparse factors (chkr_hop_half[g], chkr_hop[1], their inverses / adjoints,
// stack.jl:185-235) plus diagonal scalings, all acting on ONE index k of X (rows for a left product, columns for a
// right product).  A workgroup keeps a slab of 32 (n <= 256) or 16 values of the other index in LDS, applies the
// whole sequence there (ping-pong between two images, one barrier per factor) and writes the slab back: X moves
// through HBM once per product instead of once per factor, and no n^3 work is done.
#include "kernels.h"
#include <hip/hip_ext.h>

namespace dqmc {

// factor f, row k: val[(f * n + k) * kmax + j], col likewise (padding: val 0, col k)
template <int QW>
__global__ __launch_bounds__(256) void cb_apply_kernel(CbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double cbsm[];
    const int n = a.n, LD = QW + 1;
    double *Y0 = cbsm, *Y1 = cbsm + (size_t)n * LD;
    const int slabs = (n + QW - 1) / QW;
    const int unit = blockIdx.x / slabs, q0 = (blockIdx.x % slabs) * QW;
    const int w = unit / a.nb, blk = unit - w * a.nb;
    const double *__restrict__ X = a.X + (long)unit * a.strideX;
    double *__restrict__ O = a.O + (long)unit * a.strideX;
    const int tid = threadIdx.x;
    // scale of the mixed index k: conf-derived exp(+-lambda s) and / or a stored vector (mu)
    auto kscale = [&](int conf_sign, const double *vec, int k) -> double {
        double s = 1.0;
        if (conf_sign != 0) {
            const int8_t c = a.conf[(long)w * a.conf_stride + k];
            const double ep = blk == 0 ? a.epl : a.eml, em = blk == 0 ? a.eml : a.epl;  // block 1: opposite sign
            s = (conf_sign > 0) ? (c > 0 ? ep : em) : (c > 0 ? em : ep);
        }
        if (vec) s *= vec[(long)blk * n + k];
        return s;
    };
    // the two diagonal scalings of the mixed index, once per k
    double *spre = cbsm + 2 * (size_t)n * LD, *spost = spre + n;
    for (int k = tid; k < n; k += 256) {
        spre[k] = kscale(a.pre_conf, a.pre_vec, k);
        spost[k] = kscale(a.post_conf, a.post_vec, k);
    }
    __syncthreads();
    // load: side 0 (left product) k = row, q = column; side 1 k = column, q = row
    if (a.side == 0) {
#pragma unroll 8
        for (int idx = tid; idx < n * QW; idx += 256) {
            const int k = idx % n, q = idx / n;  // k fastest: contiguous in memory
            const int qc = min(q0 + q, n - 1);
            const double v = X[k + (long)n * qc];
            Y0[k * LD + q] = (q0 + q < n) ? v * spre[k] : 0.0;
        }
    } else {
#pragma unroll 8
        for (int idx = tid; idx < n * QW; idx += 256) {
            const int q = idx % QW, k = idx / QW;  // q fastest: QW consecutive rows
            const int qc = min(q0 + q, n - 1);
            const double v = X[qc + (long)n * k];
            Y0[k * LD + q] = (q0 + q < n) ? v * spre[k] : 0.0;
        }
    }
    __syncthreads();
    // one thread per row k of the factor: its few coefficients are loaded once per factor (coalesced over the
    // threads, requested one factor ahead), then applied to the QW slab values of that row
    constexpr int KM = 4;  // register-resident nonzeros per row (square / chain lattices need 2)
    const int rows_per_thread = (n + 255) / 256;
    for (int f = 0; f < a.seq_len; ++f) {
        const int m = a.seq[f];
        const double *__restrict__ fv = a.vals + ((size_t)m * n) * a.kmax;
        const int *__restrict__ fc = a.cols + ((size_t)m * n) * a.kmax;
        for (int i = 0; i < rows_per_thread; ++i) {
            const int k = tid + 256 * i;
            if (k >= n) break;
            if (a.kmax <= KM) {
                double cv[KM];
                int cc[KM];
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    const bool ok = j < a.kmax;
                    cv[j] = ok ? fv[k * a.kmax + j] : 0.0;
                    cc[j] = ok ? fc[k * a.kmax + j] * LD : k * LD;
                }
#pragma unroll 8
                for (int q = 0; q < QW; ++q) {
                    double sacc = cv[0] * Y0[cc[0] + q];
#pragma unroll
                    for (int j = 1; j < KM; ++j) sacc += cv[j] * Y0[cc[j] + q];
                    Y1[k * LD + q] = sacc;
                }
            } else {
                for (int q = 0; q < QW; ++q) {
                    double sacc = 0.0;
                    for (int j = 0; j < a.kmax; ++j) sacc += fv[k * a.kmax + j] * Y0[fc[k * a.kmax + j] * LD + q];
                    Y1[k * LD + q] = sacc;
                }
            }
        }
        __syncthreads();
        double *t = Y0; Y0 = Y1; Y1 = t;
    }
    if (a.side == 0) {
#pragma unroll 8
        for (int idx = tid; idx < n * QW; idx += 256) {
            const int k = idx % n, q = idx / n;
            if (q0 + q < n) {
                double v = Y0[k * LD + q] * spost[k];
                if (a.qscale) v *= a.qscale[(long)unit * a.qstride + q0 + q];
                O[k + (long)n * (q0 + q)] = v;
            }
        }
    } else {
        for (int idx = tid; idx < n * QW; idx += 256) {
            const int q = idx % QW, k = idx / QW;
            if (q0 + q < n) {
                double v = Y0[k * LD + q] * spost[k];
                if (a.qscale) v *= a.qscale[(long)unit * a.qstride + q0 + q];
                O[(q0 + q) + (long)n * k] = v;
            }
        }
    }
}

hipError_t launch_cb_apply(const CbArgs &a, int n_units, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    const int qw = a.n <= 256 ? 32 : 16;
    const size_t lds = (2 * (size_t)a.n * (qw + 1) + 2 * (size_t)a.n) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned attr_mask = 0;
    if (!(attr_mask & (1u << dev))) {
        (void)hipFuncSetAttribute((const void *)cb_apply_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)cb_apply_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_mask |= 1u << dev;
    }
    const int slabs = (a.n + qw - 1) / qw;
    if (qw == 32) hipExtLaunchKernelGGL(cb_apply_kernel<32>, dim3(n_units * slabs), dim3(256), lds, s, start, stop, 0, a);
    else hipExtLaunchKernelGGL(cb_apply_kernel<16>, dim3(n_units * slabs), dim3(256), lds, s, start, stop, 0, a);
    return hipGetLastError();
}

}  // namespace dqmc
