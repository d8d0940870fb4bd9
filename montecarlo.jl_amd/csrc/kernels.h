// kernels.h — launch interfaces of the gfx950 DQMC kernels (internal to libdqmc_hip.so).
//
// Data model: a "unit" is one n x n problem = (walker, block); unit = walker*nb + block.
// Every per-unit matrix is column-major with leading dimension ld and lives at
// base + unit*stride_unit + (unit % nb)*stride_blk  (stride_unit = 0 for the hopping
// exponentials, which are shared by all walkers and indexed by block only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dqmc {

// A length-n vector per unit that is either stored (d) or derived on the fly from
// the HS field: exp(±lambda*conf[i,l]) takes only two values
// (HubbardModelAttractive.jl:100-110, HubbardModelRepulsive.jl:113-126).
struct VecSrc {
    int mode;  // 0 = none (1.0), 1 = d[i], 2 = conf-derived, 3 = 1.0/d[i],
               // 4 = min(1, d[i]) (vmin!), 5 = 1/max(1, d[i]) (vmaxinv!, general.jl:90-117)
    const double *d;
    long stride;          // per unit
    const int8_t *conf;   // conf + walker*conf_stride + i  (already offset to the slice)
    long conf_stride;     // per walker
    double cpos[2], cneg[2];  // value for conf=+1 / conf=-1, per block
};
static inline VecSrc vs_none() { VecSrc v = {}; v.mode = 0; return v; }
static inline VecSrc vs_arr(const double *d, long stride) { VecSrc v = {}; v.mode = 1; v.d = d; v.stride = stride; return v; }
static inline VecSrc vs_inv(const double *d, long stride) { VecSrc v = {}; v.mode = 3; v.d = d; v.stride = stride; return v; }
static inline VecSrc vs_min1(const double *d, long stride) { VecSrc v = {}; v.mode = 4; v.d = d; v.stride = stride; return v; }
static inline VecSrc vs_maxinv(const double *d, long stride) { VecSrc v = {}; v.mode = 5; v.d = d; v.stride = stride; return v; }
#ifdef __HIPCC__
__device__ __forceinline__ double vs_get(const VecSrc &v, int unit, int nb, int i)
{
    if (v.mode == 1) return v.d[(long)unit * v.stride + i];
    if (v.mode == 3) return 1.0 / v.d[(long)unit * v.stride + i];
    if (v.mode == 2) {
        const int w = unit / nb, b = unit - w * nb;
        const int8_t c = v.conf[(long)w * v.conf_stride + i];
        return c > 0 ? v.cpos[b] : v.cneg[b];
    }
    if (v.mode == 4) return fmin(1.0, v.d[(long)unit * v.stride + i]);
    if (v.mode == 5) return 1.0 / fmax(1.0, v.d[(long)unit * v.stride + i]);
    return 1.0;
}
#endif

struct MatRef {
    const double *p;
    long stride_unit, stride_blk;
    int ld;
};
static inline MatRef mat(const double *p, long su, int ld, long sb = 0) { MatRef m = {p, su, sb, ld}; return m; }

// C[u] = beta*C[u] + epi( alpha * opA(A[u]) * diag(kscale) * opB(B[u]) ) + ident*I + diag(adddiag)
// epi applies colscale (index n) and rowscale (index m); row_first selects the order.
struct GemmArgs {
    int M, N, K;
    int n_units, nb;
    MatRef A, B;
    double *C; long strideC; int ldc;
    int transA, transB;
    VecSrc kscale, colscale, rowscale, adddiag;
    int row_first;
    double alpha;   // scale of the product
    double ident;   // added on the diagonal
    int beta;       // 0: overwrite, 1: accumulate into C
    // structure of the compact-WY operands (V = unit lower triangular Householder vectors with explicit zeros, M = N = K):
    //   1: C = V' V, only the 64 x 64 tiles on and above the diagonal are computed, k >= max(m0, n0) (the rest of the
    //      sum is zero; the triangle below the diagonal tiles is left untouched: nothing reads it)
    //   2: C = A V', k < n0 + 64 (V[n, k] = 0 for k > n)
    // (measured at 32 units, n = 256: no change of the launch time - 24.4 us either way: the launch lasts as long as
    // its full-k tiles, and those run at the chip's sustained fp64 MFMA rate, 45.7 TF/s = 23.5 us per 2 n^3 x 32; the
    // skipped work only saves power.  A k-tile of 32 instead of 16 did not change it either.)
    int tri;
};
// start/stop (optional): events that take the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL),
// i.e. the kernel-only duration a profiler's kernel trace reports
hipError_t launch_gemm(const GemmArgs &g, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// Chain of left products on a column slab kept in LDS (slab.hip; n = 256 only, ld = 256):
//   X_s = post_s (.) (A_s (pre_s (.) X_{s-1})), out = X_nsteps (.) colscale;  pre/post = exp(sign lambda conf[row]) or none
constexpr int SLAB_MAX_STEPS = 10;
struct SlabStep {
    const double *A; long su, sb;        // A_s at A + unit su + block sb (su = 0: shared constant)
    const int8_t *pre_conf; int pre_sign;    // conf pointers already offset to the slice; null = no scaling
    const int8_t *post_conf; int post_sign;
};
struct SlabArgs {
    int n_units, nb, nsteps;
    SlabStep st[SLAB_MAX_STEPS];
    const double *X0; long x_su, x_sb;   // X_0 (per unit, or a shared constant with x_su = 0)
    double *out; long out_su;            // must not alias X0 or any per-unit A_s
    const double *col_d; long col_stride;  // final column scaling by an array (per unit), or
    const int8_t *col_conf; int col_sign;  // by exp(sign lambda conf[col]), or none
    long conf_stride;                    // per walker
    double epl, eml;
};
hipError_t launch_slab_chain(const SlabArgs &a, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// C += U V' for the delayed-update flush: U[t + n m], VT[t + n m], m < 64 slots; n % 64 == 0
hipError_t launch_gemm_flush(int n, int n_units, const double *U, const double *VT, long sUV, double *C, long sC,
                             hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// Column-pivoted Householder QR, in place (udt_AVX_pivot! "QR decomposition" loop,
// src/linalg/UDT.jl:212-246).  On exit A holds R on/above the diagonal and the
// Householder vectors below it (unit diagonal implied), tau[n], pivot[n] (0-based:
// column j of the factored matrix is original column pivot[j]).
// Workspace of the cooperative (8 workgroups per matrix) QR: mailbox of n_units x 2 x 8 slots of
// QR_COOP_SLOT doubles (tagged packets), an error flag (bounded spins), a launch counter.
// A/B and fallback switches of the launchers (environment).  They are read ONCE per handle / primitive call
// (refresh_kernel_switches(), called when a handle is set up), never inside a launch: no environ scan per stabilisation step,
// no change of kernels or numerics in the middle of a run, no race with a host thread that edits the environment.
struct KernelSwitches {
    bool qr_stream = false;       // DQMC_QR_STREAM: streaming single-workgroup QR instead of the tile kernel
    bool qr_tile_bounds = false;  // DQMC_QR_TILE_BOUNDS: tile QR with run-time bounds at n == 256
    bool qr_nopanel = false;      // DQMC_QR_NOPANEL: streaming QR instead of the panel kernel for n > 256
    double qp_thr = 1e-4;         // DQMC_QP_THR: recompute threshold of the panel kernel's down-dated norms
    bool trsm_simple = false;     // DQMC_TRSM_SIMPLE: substitution kernel (the fallback of the MFMA solves)
    bool trsm_ll = false;         // DQMC_TRSM_LL: left-looking slab-in-LDS solve (n > 256 panels)
    bool trsm_bounds = false;     // DQMC_TRSM_BOUNDS: right-looking solve with run-time bounds at n == 256
    bool flush_ncp2 = false;      // DQMC_FLUSH_NCP2: two column passes per flush workgroup
    int gemm_stagger = 0;         // DQMC_GEMM_STAGGER
};
const KernelSwitches &kernel_switches();
void refresh_kernel_switches();

constexpr int QR_COOP_SLOT = 528;  // 264 packets of 16 bytes
constexpr int QR_COOP_SLOTS_PER_UNIT = 16;  // 2 parities x 8 parts (qr_coop_kernel)
struct QrCoopWorkspace {
    double *mailbox = nullptr;
    int *errflag = nullptr;
    int *fb = nullptr;    // [0] launch epoch of the last cooperative launch that timed out, [1] fallbacks taken
    unsigned long long epoch = 0;
    int max_blocks = 0;   // launch the cooperative kernel only if its grid fits (co-residency)
    int tail_j0 = 128;    // n == 256: hand-over step to qr_tail_kernel (0 = cooperative kernel only; 64, 96, 128)
    // A/B and test switches, read from the environment when the workspace is set up (per handle / per primitive call):
    int force_sc1 = 0;      // DQMC_QR_SC1: write-through (agent-scope) packet stores regardless of placement
    int no_coop = 0;        // DQMC_QR_NOCOOP: single-workgroup kernels only
    int force_timeout = 0;  // DQMC_QR_FORCE_TIMEOUT: 1 = every cooperative launch gives up at once;
                            // "step:<j>" -> 2 + j: part 3 of every matrix stops publishing at step j (bounded spins run out)
    // pre-pivoted blocked UDT (qrb.hip, n == 256): its own mailbox; blk_max_blocks = co-resident workgroups (0 = off)
    double *mailbox2 = nullptr;
    int blk_max_blocks = 0;
};
// udt_AVX_pivot! (UDT.jl:192-306) at n == 256 in one launch: U, D, T = D^-1 R (pivot applied or not, out of place), pivot
// (0-based positions -> original columns).  Pivot order = descending norm of the input columns (qrb.hip).
// B (optional, n_units x strideB, must not alias U): U receives B Q instead of Q (the product the reference forms right behind
// the decomposition in calculate_greens_AVX!, stack.jl:360 / :378)
hipError_t launch_udt_blocked(int n_units, const double *A, long strideA, double *U, long strideU, double *D, long strideD,
                              double *T, long strideT, int *pivot, QrCoopWorkspace *ws, int apply_pivot, hipStream_t s,
                              const double *B = nullptr, long strideB = 0);
size_t qrb_mailbox_bytes(int n_units);
int qrb_blocks_per_cu();
// ws may be null: single-workgroup kernels only
// W (n_units x strideW, may be null): output of the cooperative kernel, which leaves A intact so that a time-out of
// its hand-offs can be recovered by the single-workgroup kernel launched (guarded) behind it; *factored tells where
// the factored matrix is (W or A)
// X (n_units x strideX, may be null): hand-over buffer of the two-phase factorisation at n == 256 (cooperative steps
// 0..63, then one CU per matrix, qr_tail_kernel); its contents are scratch
hipError_t launch_qr_pivot(int n, int n_units, double *A, long strideA, double *tau, int *pivot,
                           QrCoopWorkspace *ws, double *W, long strideW, const double **factored, hipStream_t s,
                           double *X = nullptr, long strideX = 0);
int qr_coop_blocks_per_cu();

// After launch_qr_pivot: D = |diag R| (UDT.jl:268-272); V = unit-lower Householder
// vectors (n x n, explicit zeros/ones); T = D^-1 R:
//   apply_pivot = 1: Tout[i, pivot[j]] = R[i,j]/D[i], zero elsewhere (UDT.jl:283-297), out of place
//   apply_pivot = 0: A[i,j] *= 1/D[i] for j >= i in place, below-diagonal left dirty (UDT.jl:298-306)
// F = the factored matrix (A itself for the in-place kernels)
hipError_t launch_udt_finish(int n, int n_units, double *A, long strideA, const double *F, long strideF,
                             const int *pivot, double *D, long strideD, double *V, long strideV, double *Tout,
                             long strideT, int apply_pivot, hipStream_t s);

// X = gather(A)[:, pivot] * triu(T)^-1 written to Out (rdivp!, src/linalg/general.jl:138-166).
// pivot may be null (identity).  If dmul != null the diagonal of T is ignored and
// column j is multiplied by dmul[j] instead of divided by T[j,j] (used with the
// compact-WY triangle, whose inverse diagonal is tau).
hipError_t launch_trsm_right_upper(int n, int n_units, const double *A, long strideA, const double *T,
                                   long strideT, const int *pivot, const double *dmul, long strideV,
                                   double *Out, long strideOut, double *winv, hipStream_t s,
                                   double *scratch = nullptr);
// winv: n_units * ceil(n/16) * 256 doubles of scratch for the inverted diagonal blocks (n <= 256 path);
// nullptr selects the substitution kernel; scratch: n_units x strideOut doubles for the panelled MFMA solve of n > 256
// (without it the substitution kernel is used there)

// One chunk of sweep_spatial (DQMC.jl:546-582): sites [site0, site0+nsites) of the
// current slice with delayed rank-1 updates; writes the accepted update vectors
// (Uout: n x KD, VTout: n x KD, zero padded) for the flush GEMM G += Uout*VTout'.
struct SweepConsts {
    // per conf value index ci = (conf>0): attractive gamma (Attractive.jl:121) and
    // exp(-dE_boson); repulsive Delta_up, Delta_dn (Repulsive.jl:139-141)
    double gamma[2], ebos[2], dup[2], ddn[2];
};
struct WalkerRng {           // device-resident, one per walker
    unsigned long long seed;
    unsigned long long draw;       // draws consumed (philox counter or array cursor)
    const double *uniforms;        // non-null: host-supplied stream
    unsigned long long n_uniforms;
    int exhausted;
};
struct DevMagStats { double max, min, sum; long long count; };
struct DevStats {
    long long prop_local, acc_local;
    DevMagStats negative_probability, propagation_error;
};
#ifdef __HIPCC__
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
#endif
int sweep_kd(int n, int nb);  // chunk length (update slots per flush) for this problem size
hipError_t launch_sweep_chunk(int n, int nb, int n_walkers, int model, double *G, long strideG,
                              int8_t *conf_slice, long conf_stride, int site0, int nsites,
                              double *Uout, double *VTout, long strideUV, SweepConsts sc,
                              WalkerRng *rng, DevStats *stats, int check_sign, hipStream_t s);

// The same chunk as a decide / apply pair (sweep_lu.hip): sweep_lu_kernel eliminates the 64 x 64 block G[c, c]
// with one wave per walker (decisions, HS field, counters) and leaves register images of the triangular factors
// (sweep_lu_image_doubles() doubles per unit); sweep_flush_lu_kernel applies the chunk to G out of place
// (Gout = Gin + T R0).  No limit on n_blocks * n_sites.
size_t sweep_lu_image_doubles();
hipError_t launch_sweep_lu(int n, int nb, int n_walkers, const double *G, long strideG, int8_t *conf_slice,
                           long conf_stride, int site0, int nsites, double *img, SweepConsts sc, WalkerRng *rng,
                           DevStats *stats, int check_sign, int *errflag, hipStream_t s,
                           hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// elimination of chunk `site0` beside the flush of the previous chunk `site0p` (one launch; n % 64 == 0)
hipError_t launch_sweep_fused(int n, int nb, int n_walkers, const double *Gin, double *Gout, long strideG,
                              int8_t *conf_slice, long conf_stride, int site0, int site0p, double *img,
                              const double *imgp, SweepConsts sc, WalkerRng *rng, DevStats *stats, int check_sign,
                              int *errflag, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

hipError_t launch_sweep_flush_lu(int n, int n_units, const double *Gin, double *Gout, long strideG, int site0,
                                 int nsites, const double *img, hipStream_t s, hipEvent_t start = nullptr,
                                 hipEvent_t stop = nullptr);

// Checkerboard products with sparse bond-group factors (cb.hip): O = post . F_last ... F_first . pre . X on the rows
// (side 0) or columns (side 1) of X; factor m in ELL form vals / cols [m][n][kmax]; the diagonal scalings are the
// conf-derived exp(+-lambda s) (sign +1 / -1, 0 = none) and / or a stored vector per block (mu); qscale scales the
// other index (the Diagonal(D) of add_slice_sequence_left/right).
struct CbArgs {
    int n, nb, side, kmax, seq_len;
    int seq[32];
    const double *vals;
    const int *cols;
    const double *X;
    double *O;
    long strideX;
    const int8_t *conf;   // already offset to the slice
    long conf_stride;
    double epl, eml;
    int pre_conf, post_conf;
    const double *pre_vec, *post_vec;   // [nb][n] or null
    const double *qscale;               // [units][qstride] or null
    long qstride;
};
hipError_t launch_cb_apply(const CbArgs &a, int n_units, hipStream_t s, hipEvent_t start = nullptr,
                           hipEvent_t stop = nullptr);

// small helpers
hipError_t launch_set_identity(int n, int count, double *A, long stride, hipStream_t s);
hipError_t launch_fill(double *p, size_t n, double v, hipStream_t s);
// elementwise helpers of the unequal-time path: A = Diagonal(d) (copyto!(A, Diagonal(d))), A += B (rvadd!),
// O = A - I (vsub!, general.jl:67-85), d = f(d) in place with f given by the VecSrc mode
hipError_t launch_set_diag(int n, int nb, int units, double *A, long stride, VecSrc d, hipStream_t s);
hipError_t launch_mat_add(double *A, const double *B, size_t count, hipStream_t s);
hipError_t launch_sub_identity(int n, int units, double *O, const double *A, long stride, hipStream_t s);
hipError_t launch_vec_map(int n, int nb, int units, double *dst, long stride, VecSrc src, hipStream_t s);
hipError_t launch_scale_mat(int n, int nb, int units, double *O, const double *A, long stride, VecSrc row, VecSrc col,
                            int row_first, hipStream_t s);
// per walker max|A-B| over its nb blocks; pushes log10 into stats.propagation_error if > 1e-7
// scratch: 2 zero-initialised words per walker (partial maximum, arrival counter; left zeroed by every launch)
hipError_t launch_prop_check(int n, int nb, int n_walkers, const double *A, const double *B,
                             long stride_unit, DevStats *stats, unsigned long long *scratch, hipStream_t s);
// acc += sum over walkers of G, G.^2, 1-diag(G); layout documented in include/dqmc_hip.h
hipError_t launch_accumulate(int n, int nb, int n_walkers, const double *G, long stride_unit,
                             double *acc, hipStream_t s);
hipError_t launch_mfma_peak(int iters, int blocks, double *sink, hipStream_t s);
// equal-time correlations (cdc, sdc_x/y/z per direction; mx, my, mz per site), summed over walkers:
// acc layout [cdc nd][sdc_x nd][sdc_y nd][sdc_z nd][mx n][my n][mz n][count]
hipError_t launch_correlations(int n, int nb, int model, int n_walkers, const double *G, long stride_unit,
                               const int *dir_ptr, const int *pair_src, const int *pair_trg, int n_dirs,
                               double *per_walker, double *acc, hipStream_t s);
// pc_kernel over EachLocalQuadByDistance{K}: trg_of[src + n*k] = target of src in direction k (< K), -1 if none;
// per_walker [walkers][n_dirs*K*K], acc [n_dirs*K*K + 1] (Julia layout [dir12, dir1, dir2], then the sample count)
hipError_t launch_pairing(int n, int nb, int n_walkers, const double *G, long stride_unit, const int *dir_ptr,
                          const int *pair_src, const int *pair_trg, int n_dirs, int K, const int *trg_of,
                          double *per_walker, double *acc, hipStream_t s);
// susceptibilities: one time slice of the packed kernels added to per_walker [walkers][4*n_dirs (+ n_dirs*K*K)]
// ([cds][sds_x][sds_y][sds_z][ps]); the reduce multiplies by delta_tau and adds the sample count
hipError_t launch_sus_slice(int n, int nb, int model, int n_walkers, const double *G00, const double *G0l,
                            const double *Gl0, const double *Gll, long stride_unit, const int *dir_ptr,
                            const int *pair_src, const int *pair_trg, int n_dirs, int K, const int *trg_of,
                            double *per_walker, long per_stride, hipStream_t s);
hipError_t launch_sus_reduce(int n_walkers, long total, double factor, const double *per_walker, double *acc,
                             hipStream_t s);
// HS field <-> Julia BitArray chunks (compress / decompress, HubbardModel.jl:56-59)
hipError_t launch_conf_pack(const int8_t *conf, size_t n_elem, unsigned long long *chunks, hipStream_t s);
hipError_t launch_conf_unpack(const unsigned long long *chunks, size_t n_elem, int8_t *conf, hipStream_t s);

}  // namespace dqmc
