// qrb.hip — udt_AVX_pivot! (src/linalg/UDT.jl:192-306) for n = 256 as ONE launch: a blocked Householder QR whose pivot
// order is fixed up front (round 4).
//
// What it replaces: the reference's loop (UDT.jl:212-246) searches the largest trailing column at each of its n steps; on
// the device that was 256 dependent steps with two inter-workgroup hand-offs each (qr_coop_kernel + qr_tail_kernel), followed by
// five more launches for D, T and Q (udt_finish, V'V, TRSM, Q GEMM).  Here the column order is taken ONCE, from the norms of the
// input (descending, first maximum first - the order the reference's first step would start from), and then
//   * the 256 sorted columns are 8 panels of 32; workgroup w of the matrix OWNS panel w (owner computes: no column ever moves),
//   * a panel is factored inside its workgroup with no hand-off at all (one LDS-only barrier per step), while a fifth wave
//     scales the reflectors, accumulates the panel's compact-WY triangle T_w column by column (dlarft recurrence; the dot products
//     V[:,0:j]' v_j it needs fall out of the step for free: they are what the FINISHED columns' lanes compute anyway) and
//     publishes V_w, T_w and diag R as self-validating tagged granules,
//   * every other workgroup applies the block reflector (I - V T' V') to its own columns with v_mfma_f64_16x16x4_f64, the columns
//     living in MFMA ACCUMULATOR layout for the whole kernel: an accumulator register is at the same time the B operand of
//     W = V' C, so nothing moves between lanes (C/D: row = (lane>>4) + 4 reg, col = lane & 15; B: k = lane>>4),
//   * Q is accumulated on the way: every workgroup also carries Z = its 32 columns of Q' = H_7' ... H_0' E through the same
//     block reflectors (in the time it would otherwise wait for the next panel), so U = Q costs no pass of its own,
//   * D = |diag R| and T = D^-1 R (pivot applied or not, UDT.jl:283-306) are written by the panel's owner.
// 8 hand-offs per factorisation instead of 2 x 128 + 128 steps, one launch instead of eight.
//
// Same Householder vectors, tau, R as an unblocked factorisation of the pre-sorted matrix (compact-WY is a reassociation);
// the reference's own column order differs (it re-evaluates the norms at every step).  G is unaffected to ~1e-12 at config 3
// (measured with the CPU checker's pre-sort study switch); parity is asserted on G / the HS field against the
// reference-rule oracle and on U, D, T against the oracle run with the same pre-sorted order.
//
// Hand-offs: granule = 16 bytes {low half, tag, high half, tag}, written / read as two 8-byte relaxed atomics, tag = launch
// epoch (the mailbox is never cleared).  A reader that sees both tags has the payload: no flag, no fence.  Stores are
// workgroup-scope (they stay in the XCD's L2) only after the eight workgroups have verified at run time that they share one XCD
// (HW_REG_XCC_ID exchanged through agent-scope granules); otherwise agent-scope (write-through).  Loads are always agent-scope
// (L1-bypassing).  Every spin is bounded; a time-out raises errflag bit 4 (the caller's results are invalid and it is told so).
// A workgroup only ever waits for LOWER parts of its own matrix, so with in-order dispatch the wait always ends.
#include "kernels.h"

namespace dqmc {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned long long qword;

constexpr int QB_THREADS = 256;  // one wave per SIMD: 512 registers per lane
constexpr int VLD = 260;         // LDS column stride of a staged panel (doubles): (nn, kq) operand reads spread over the banks
constexpr int TLD = 36;
constexpr int UST = 34;          // stride of the per-row-group slices of a step's reflector (conflict-free 16-byte broadcasts)
constexpr unsigned QB_SPIN = 1000000u;  // ~1 s of polling: far beyond any legitimate wait (a whole factorisation is ~0.2 ms)

// mailbox of one unit, in granules
constexpr long MB_V = 0;                    // [panel 8][column 32][row 256]
constexpr long MB_T = MB_V + 8L * 32 * 256;   // [panel 8][column 32][row 32]
constexpr long MB_D = MB_T + 8L * 32 * 32;    // [256] diag R
constexpr long MB_N = MB_D + 256;           // [256] squared column norms of the input
constexpr long MB_X = MB_N + 256;           // [8] XCC id of each part
constexpr long MB_P = MB_X + 8;             // [256] position of every input column in the sorted order
constexpr long MB_GRANULES = MB_P + 256;

#define QB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// the workgroup's abort word in LDS: relaxed atomic accesses (re-read every time, no waits attached)
__device__ __forceinline__ int abort_get(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void abort_set(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <bool LOCAL>
__device__ __forceinline__ void g_put(qword *g, double v, unsigned tag)
{
    const qword bits = (qword)__double_as_longlong(v);
    const qword lo = (bits & 0xffffffffull) | ((qword)tag << 32), hi = (bits >> 32) | ((qword)tag << 32);
    if (LOCAL) {
        __hip_atomic_store(g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(g + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_store(g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void g_put_sel(qword *g, double v, unsigned tag, bool local)
{
    if (local) g_put<true>(g, v, tag);
    else g_put<false>(g, v, tag);
}
__device__ __forceinline__ bool g_try(const qword *g, unsigned tag, double &v)
{
    const qword a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const qword b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}
// bounded wait for one granule
__device__ __forceinline__ double g_wait(const qword *g, unsigned tag, int *s_abort)
{
    double v = 0.0;
    for (unsigned s = 0; s < QB_SPIN; ++s) {
        if (g_try(g, tag, v)) return v;
        if ((s & 63u) == 63u && abort_get(s_abort)) return 0.0;
        __builtin_amdgcn_s_sleep(2);
    }
    abort_set(s_abort, 1);
    return 0.0;
}
// NJ granules at a fixed stride, all requests in flight at once; repeated until every tag is right
template <int NJ>
__device__ __forceinline__ void g_batch(const qword *base, long stride_q, unsigned tag, double (&v)[NJ], int *s_abort)
{
    for (unsigned s = 0; s < QB_SPIN; ++s) {
        qword a[NJ], b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            a[j] = __hip_atomic_load(base + j * stride_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b[j] = __hip_atomic_load(base + j * stride_q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ok = ok & ((unsigned)(a[j] >> 32) == tag) & ((unsigned)(b[j] >> 32) == tag);
            v[j] = __longlong_as_double((long long)((a[j] & 0xffffffffull) | (b[j] << 32)));
        }
        if (ok) return;
        if ((s & 63u) == 63u && abort_get(s_abort)) return;
        __builtin_amdgcn_s_sleep(2);
    }
    abort_set(s_abort, 1);
}

template <int CTRL>
__device__ __forceinline__ double qdpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 8 lanes {8g .. 8g+7}; every lane receives the total
__device__ __forceinline__ double qsum8(double x)
{
    x += qdpp<0xB1>(x);   // quad_perm [1,0,3,2]
    x += qdpp<0x4E>(x);   // quad_perm [2,3,0,1]
    x += qdpp<0x141>(x);  // row_half_mirror
    return x;
}
// sqrt(x) and 1/sqrt(x) together (coupled Goldschmidt iterations from v_rsq_f64, then one correction of the root)
__device__ __forceinline__ void q_sqrt_rsqrt(double x, double &root, double &rroot)
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
__device__ __forceinline__ double q_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

#ifdef QRB_STAMPS  // diagnostic build only (make stamps; tools/qrb_stamps.py): s_memtime of thread 0 of every part of unit 0
__device__ long long *qrb_stamp_ptr = nullptr;  // [part 8][event 64]
#define QRB_STAMP(ID)                                                                 \
    do {                                                                              \
        if (L.stamps && (threadIdx.x == 0)) L.stamps[ID] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define QRB_STAMP(ID)
#endif
#ifdef QRB_FINE  // diagnostic: cycle stamps inside the steps of one part's panel (tools/qrb_fine.py)
__device__ long long *qrb_fine_ptr = nullptr;  // [wave 4][step 32][point 8]
#define QRB_FINE_PT(P)                                                                                   \
    do {                                                                                                 \
        if (W == QRB_FINE && lane == 0) L.fst[(wv * 32 + j) * 8 + (P)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define QRB_FINE_PT(P)
#endif
#if defined(QRB_X_DUMP) || defined(QRB_X_DUMP2)
__device__ double qrb_dump[256 * 32];
#endif
struct QbLds {
    double *Vs;      // [32][VLD] staged panel V_p (column-major, natural rows); also the layout-conversion buffer
    double *Ts;      // [32][TLD] T_p, column-major
    double *Wx;      // [4 waves][16][64] partial W = V'[C Z] of each wave, met by its row-half partner
    double *ub;      // [2][8][UST] the step's unscaled reflector u = xi v, sliced by row group
    double *scal;    // [2][2] {1/xi, tau/xi} of the step
    double *dotbuf;  // [2][32] u_j' x_c of every panel column c
    double *Th;      // [32][UST] T-hat = T diag(1/xi) of my panel, row-major
    double *rcps;    // [32] 1/xi, [32] tau, [32] nu of my panel's steps
    double *dinv;    // [256] 1/D of the rows whose panel has been seen
    double *dval;    // [32] D of my panel
    double *nrm;     // [256] input column norms
    int *ord;        // [256] position -> original column
    int *xcc;        // [8]
    int *s_abort;  // (never volatile: a volatile access makes the compiler wait for every store in flight, vmcnt(0))
    long long *stamps;  // diagnostic build: event times of this part (unit 0), else null
    long long *fst;     // diagnostic build (QRB_FINE): [wave 4][step 32][point 8]
};

// ---- panel p from the mailbox into LDS: one poll on the granule its owner publishes last, then everything in one batch ----
// HALF = 0 / 1: only reflectors 16 HALF .. 16 HALF + 15 with the diagonal block of T that belongs to them (the panel's block
// reflector is the product of its two halves' block reflectors: the next owner applies the first half while the second is
// still being factored)
// POLL = false: no separate poll - the batch itself is repeated until every tag is current.  For the hand-off the next owner
// waits for (one trip to the L2 less between the owner's last store and the first MFMA: 1.4 us per hand-off) and for panels that
// were published long ago; the other waits keep the light poll on one granule (a waiting workgroup that repeated the whole
// batch would put 64-128 KB per pass on the L2 the owners publish through)
template <int P, int HALF = -1, bool POLL = true>
__device__ __forceinline__ void qrb_fetch(const qword *mb, unsigned tag, const QbLds &L, int tid)
{
    constexpr int C0 = HALF < 0 ? 0 : 16 * HALF, NC = HALF < 0 ? 32 : 16;  // reflectors C0 .. C0 + NC - 1
    const int r = tid;
    // uniform column bases + one 32-bit lane offset: the requests of a pass then share ONE address register (their
    // column offsets do not fit the instruction's immediate, and 64 address pairs cost 128 registers)
    const qword *vb = mb + 2 * (MB_V + ((long)P * 32) * 256);
    const unsigned ro = 2u * (unsigned)r;
    const qword *tb = mb + 2 * (MB_T + ((long)P * 32) * 32);
    // my share of T: full panel: column tid >> 3, rows 4 (tid & 7) ..+3; half: entry (row C0 + (tid & 15), column C0 + (tid >> 4))
    constexpr int NT = HALF < 0 ? 4 : 1;
    const int tcol = HALF < 0 ? (tid >> 3) : C0 + (tid >> 4), trow = HALF < 0 ? 4 * (tid & 7) : C0 + (tid & 15);
    const qword *tmine = tb + 2L * (tcol * 32 + trow);
    const qword *dmine = mb + 2 * (MB_D + 32 * P + C0 + (tid & (NC - 1)));
    QRB_STAMP(8 + 4 * P + 0);
    if (POLL) (void)g_wait(tb + 2L * ((C0 + NC - 1) * 32 + C0 + NC - 1), tag, L.s_abort);
    QRB_STAMP(8 + 4 * P + 1);
    for (unsigned s = 0; s < QB_SPIN; ++s) {
        bool ok = true;
        // (values go to LDS as they come; a pass with a stale granule is simply repeated)
        if (r >= 32 * P) {
#pragma unroll
            for (int half = 0; half < NC / 16; ++half) {
                qword a[16], b[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const qword *cb = vb + 2L * (C0 + 16 * half + j) * 256;
                    a[j] = __hip_atomic_load(cb + ro, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b[j] = __hip_atomic_load(cb + ro + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    ok = ok & ((unsigned)(a[j] >> 32) == tag) & ((unsigned)(b[j] >> 32) == tag);
                    L.Vs[(C0 + 16 * half + j) * VLD + r] = __longlong_as_double((long long)((a[j] & 0xffffffffull) | (b[j] << 32)));
                }
            }
        }
        {
            qword a[NT + 1], b[NT + 1];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                a[j] = __hip_atomic_load(tmine + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[j] = __hip_atomic_load(tmine + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a[NT] = __hip_atomic_load(dmine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b[NT] = __hip_atomic_load(dmine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < NT + 1; ++j) ok = ok & ((unsigned)(a[j] >> 32) == tag) & ((unsigned)(b[j] >> 32) == tag);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                L.Ts[tcol * TLD + trow + j] = __longlong_as_double((long long)((a[j] & 0xffffffffull) | (b[j] << 32)));
            if (tid < NC)
                L.dinv[32 * P + C0 + tid] = 1.0 / fabs(__longlong_as_double((long long)((a[NT] & 0xffffffffull) | (b[NT] << 32))));
        }
        if (ok) break;
        if (((s & 63u) == 63u && abort_get(L.s_abort)) || s + 1 == QB_SPIN) {
            abort_set(L.s_abort, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    QB_BARRIER();
    QRB_STAMP(8 + 4 * P + 2);
}

// ---- [C Z] <- (I - V_p T_p' V_p') [C Z] on the row tiles t >= 2P ---------------------------------------------------
template <int P, bool DO_C, bool DO_Z, int HALF = -1>
__device__ __forceinline__ void qrb_apply(d4 (&c)[8], d4 (&z)[8], const QbLds &L, int wv, int lane)
{
    constexpr int A0 = HALF < 0 ? 0 : HALF, A1 = HALF < 0 ? 2 : HALF + 1;  // tiles of 16 reflectors that take part
#ifdef QRB_X_NOAPPLY
    return;
#endif
    const int nn = lane & 15, kq = lane >> 4, rh = wv >> 1;
    d4 wc[2], wz[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        wc[a] = (d4){0.0, 0.0, 0.0, 0.0};
        wz[a] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    // W = V' [C Z], my row tiles only
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        const double *va = L.Vs + (16 * a + nn) * VLD + kq;
#pragma unroll
        for (int u = P; u < 8; ++u) {
            const int t = 2 * u + rh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = va[16 * t + 4 * r];
                if (DO_C) wc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, c[u][r], wc[a], 0, 0, 0);
                if (DO_Z) wz[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, z[u][r], wz[a], 0, 0, 0);
            }
        }
    }
    // the two row halves of a column tile meet
    {
        double *mine = L.Wx + (wv * 16) * 64 + lane;
#pragma unroll
        for (int a = A0; a < A1; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (DO_C) mine[(a * 4 + r) * 64] = wc[a][r];
                if (DO_Z) mine[(8 + a * 4 + r) * 64] = wz[a][r];
            }
        QB_BARRIER();
        const double *other = L.Wx + ((wv ^ 2) * 16) * 64 + lane;
#pragma unroll
        for (int a = A0; a < A1; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (DO_C) wc[a][r] += other[(a * 4 + r) * 64];
                if (DO_Z) wz[a][r] += other[(8 + a * 4 + r) * 64];
            }
    }
    // Y = -T' W (T upper triangular: block (a, a') only for a <= a')
    d4 yc[2], yz[2];
#pragma unroll
    for (int ap = A0; ap < A1; ++ap) {
        yc[ap] = (d4){0.0, 0.0, 0.0, 0.0};
        yz[ap] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = A0; a <= ap; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double tv = -L.Ts[(16 * ap + nn) * TLD + 16 * a + 4 * r + kq];
                if (DO_C) yc[ap] = __builtin_amdgcn_mfma_f64_16x16x4f64(tv, wc[a][r], yc[ap], 0, 0, 0);
                if (DO_Z) yz[ap] = __builtin_amdgcn_mfma_f64_16x16x4f64(tv, wz[a][r], yz[ap], 0, 0, 0);
            }
    }
    // [C Z] += V Y
#pragma unroll
    for (int u = P; u < 8; ++u) {
        const int t = 2 * u + rh;
#pragma unroll
        for (int ap = A0; ap < A1; ++ap)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = L.Vs[(16 * ap + 4 * r + kq) * VLD + 16 * t + nn];
                if (DO_C) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, yc[ap][r], c[u], 0, 0, 0);
                if (DO_Z) z[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, yz[ap][r], z[u], 0, 0, 0);
            }
    }
    QB_BARRIER();  // Vs, Ts, Wx are free again
    QRB_STAMP(8 + 4 * P + 3);
}

// column jj of the panel's compact-WY triangle (dlarft, forward / columnwise): T[0:jj, jj] = -tau T[0:jj, 0:jj] V[:, 0:jj]' v_jj.
// dots[k] = u_k' u_jj for the finished columns k < jj (entries k >= jj are other columns' dot products: they meet zeros of
// T-hat); T-hat = T diag(1/xi) stays in LDS.  One wave, lanes 0..31 = rows; all requests first, then the sums.
template <int W>
__device__ __forceinline__ void qrb_tcolumn(const QbLds &L, qword *mb, unsigned tag, bool local, int lane, int jj, bool publish = true)
{
    // lanes (row c = lane & 31, half h = lane >> 5): half h sums over k in [16 h, 16 h + 16); the halves meet with one
    // v_permlane32_swap (the 48 LDS reads of a 32-lane form made this the longest job of a step: LDS passes, not arithmetic)
    const int cl = lane & 31, h = lane >> 5;
    const double2 *dots = reinterpret_cast<const double2 *>(L.dotbuf + (jj & 1) * 32 + 16 * h);
    const double2 *th = reinterpret_cast<const double2 *>(L.Th + cl * UST + 16 * h);
    double2 dd[8], tt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dd[k] = dots[k];
        tt[k] = th[k];
    }
    const double tau = L.rcps[32 + 2 * jj], rcp = L.rcps[jj];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a0 = __builtin_fma(tt[k].x, dd[k].x, a0);
        a1 = __builtin_fma(tt[k].y, dd[k].y, a1);
    }
    const double part = a0 + a1;
    const int plo = __double2loint(part), phi = __double2hiint(part);
    const auto slo = __builtin_amdgcn_permlane32_swap(plo, plo, false, false);
    const auto shi = __builtin_amdgcn_permlane32_swap(phi, phi, false, false);
    const double acc = part + __hiloint2double((int)shi[1], (int)slo[1]);  // (lanes 0..31: + the sum of lanes 32..63)
    if (lane < 32) {
        const double t = lane < jj ? -(tau * rcp) * acc : (lane == jj ? tau : 0.0);
        L.Th[lane * UST + jj] = t * rcp;
        if (publish && abort_get(L.s_abort) != 2) g_put_sel(mb + 2 * (MB_T + ((long)W * 32 + jj) * 32 + lane), t, tag, local);
    }
}

__device__ __forceinline__ double qmovdpp_b1(double x)  // quad_perm [1,0,3,2]
{
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ double qmovdpp_4e(double x)  // quad_perm [2,3,0,1]
{
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ double qmovdpp_hm(double x)  // row_half_mirror
{
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0x141, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(x), 0x141, 0xf, 0xf, true));
}
// sum over the 8 lanes {8g .. 8g+7} (every source lane is valid: no old value to prepare)
__device__ __forceinline__ double qsum8b(double x)
{
    x += qmovdpp_b1(x);
    x += qmovdpp_4e(x);
    x += qmovdpp_hm(x);
    return x;
}

// ---- panel W factored by its owner -------------------------------------------------------------------------------
// thread (pc = tid >> 3, rg = tid & 7) holds column pc of the panel, rows 16 (k >> 1) + 2 rg + (k & 1), k = 0..31 (rows below
// 32 W carry finished R entries).  The 32 steps are four ERAS of eight: in era E wave E owns the pivot columns, and every
// wave runs straight-line code for its role of the era (taken branches cost ~50 cycles each on a lone wave - measured with
// in-kernel stamps - so roles, register choices and buffer parities are template constants, not run-time tests):
//   role 0 (wave E)      [A](j): norm over the 8 lanes of the column, nu / xi / beta = 1 / (nu xi) (UDT.jl:133-148 with
//                        H = I - beta u u', u = xi v: no scaling pass), diagonal entry patched into the reflector that has
//                        been in LDS since the end of the previous step; then, like everybody, ONE barrier and
//                        [B](j): dot with u, update (finished columns with coefficient 0: their dot products are V' v_j),
//                        the finished row g leaves x for r, and the next column goes to LDS as the next raw reflector;
//   role 1 (wave E + 1)  [B], then v_j = u / xi to the mailbox (at the era's last step: the raw reflector of ITS first column);
//   role 2 (wave E + 2)  [B], then column j - 1 of T (dlarft recurrence);
//   role 3 (wave E + 3)  [B], then diag R and 1 / D (at the era's last step also v_j, for role 1).
struct QbStepCtx {
    qword *mb;
    unsigned tag;
    bool local, real;
    int lane, pc, rg, stop_at;
};

template <int W>
__device__ __forceinline__ void qrb_publish_v(const QbLds &L, const QbStepCtx &C, int par, int j)
{
    constexpr int NI = (256 - 32 * W + 63) / 64;
    const int g = 32 * W + j;
    const double xi = L.scal[par * 2 + 1];
    double u[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = min(32 * W + C.lane + 64 * i, 255);
        u[i] = L.ub[(par * 8 + ((row & 15) >> 1)) * UST + 2 * (row >> 4) + (row & 1)];
    }
    const double rcp = q_rcp(xi);
    const bool quiet = abort_get(L.s_abort) == 2 || !C.real;  // forced time-out: nothing is published any more
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = 32 * W + C.lane + 64 * i;
        if (row < 256) {
            const double v = row > g ? u[i] * rcp : (row == g ? 1.0 : 0.0);
            if (!quiet) g_put_sel(C.mb + 2 * (MB_V + ((long)W * 32 + j) * 256 + row), v, C.tag, C.local);
        }
    }
    if (C.lane == 0) L.rcps[j] = rcp;
}

template <int W, int E, int ROLE, int ODD>
__device__ __forceinline__ void qrb_step(double (&x)[32], double (&r)[4], const QbLds &L, const QbStepCtx &C, int j)
{
    constexpr int KB0 = 4 * W, JB = E >> 1, KK = KB0 + 2 * JB + ODD;  // x[KK] holds row g in the lane with rg == (j & 15) >> 1
    const int g = 32 * W + j, rgo = (j & 15) >> 1;
    const bool last = (j & 7) == 7;  // last step of the era
    double *ug = L.ub + (ODD * 8 + rgo) * UST + KK;  // LDS word of row g in the reflector buffer
    if (j == C.stop_at && C.lane == 0) abort_set(L.s_abort, 2);  // test hook: this part stops publishing
    if (ROLE == 0) {
        // every column group of the wave runs the same chain on its own norm; only the owner's results are stored
        const double xi1 = *ug;  // (broadcast read; the raw column is there since the end of the previous step)
        double n0 = 0.0, n1 = 0.0;
#pragma unroll
        for (int k = KB0; k < 32; k += 2) {
            n0 = __builtin_fma(x[k], x[k], n0);
            n1 = __builtin_fma(x[k + 1], x[k + 1], n1);
        }
        const double maxval = qsum8b(n0 + n1);
        // UDT.jl:133-148, branch-free: a zero column keeps tau = 0 and stays as it is (D = 0, as in the reference)
        const bool nz = maxval != 0.0;
        double rootn, rrootn;
        q_sqrt_rsqrt(nz ? maxval : 1.0, rootn, rrootn);
        const double nu = nz ? copysign(rootn, xi1) : -xi1;
        const double xi = nz ? xi1 + nu : 1.0;
        const double tj = nz ? __builtin_fma(fabs(xi1), rrootn, 1.0) : 0.0;  // xi / nu
        const double beta = nz ? q_rcp(nu * xi) : 0.0;                       // tau / xi^2
        if (C.lane == 8 * (j & 7)) {
            *ug = xi;
            *reinterpret_cast<double2 *>(L.scal + ODD * 2) = make_double2(beta, xi);
            *reinterpret_cast<double2 *>(L.rcps + 32 + 2 * j) = make_double2(tj, nu);
        }
    }
    QB_BARRIER();
    {
        const double beta = L.scal[ODD * 2];
        const double2 *uq = reinterpret_cast<const double2 *>(L.ub + (ODD * 8 + C.rg) * UST);
        double2 uu[16];
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = KB0; k < 32; k += 2) {
            uu[k >> 1] = uq[k >> 1];
            d0 = __builtin_fma(uu[k >> 1].x, x[k], d0);
            d1 = __builtin_fma(uu[k >> 1].y, x[k + 1], d1);
        }
        const double dot = qsum8b(d0 + d1);
        if (C.rg == 0) L.dotbuf[ODD * 32 + C.pc] = dot;
        const bool live = C.pc > j;
        const double coef = live ? dot * beta : 0.0;
#pragma unroll
        for (int k = KB0; k < 32; k += 2) {
            x[k] = __builtin_fma(-uu[k >> 1].x, coef, x[k]);
            x[k + 1] = __builtin_fma(-uu[k >> 1].y, coef, x[k + 1]);
        }
        // row g of the live columns is finished: it leaves x (one lane per column)
        const bool mine = live & (C.rg == rgo);
        r[2 * JB + ODD] = mine ? x[KK] : r[2 * JB + ODD];
        x[KK] = mine ? 0.0 : x[KK];
    }
    if (ROLE == 0) {
        // the next column's raw reflector goes to LDS at once (its diagonal entry is patched when xi is known)
        if (!last && (C.lane >> 3) == ((j + 1) & 7)) {
            double2 *dst = reinterpret_cast<double2 *>(L.ub + ((ODD ^ 1) * 8 + C.rg) * UST);
#pragma unroll
            for (int k = KB0; k < 32; k += 2) dst[k >> 1] = make_double2(x[k], x[k + 1]);
        }
    } else if (ROLE == 1) {
        if (last && E < 3) {  // my first column is the next pivot column
            if (C.lane < 8) {
                double2 *dst = reinterpret_cast<double2 *>(L.ub + ((ODD ^ 1) * 8 + C.rg) * UST);
#pragma unroll
                for (int k = KB0; k < 32; k += 2) dst[k >> 1] = make_double2(x[k], x[k + 1]);
            }
        } else {
            qrb_publish_v<W>(L, C, ODD, j);
        }
    } else if (ROLE == 2) {
        if (j >= 1) qrb_tcolumn<W>(L, C.mb, C.tag, C.local, C.lane, j - 1, C.real);
    } else {
        if (C.lane == 0) {  // diag R and 1 / D
            const double nu = L.rcps[32 + 2 * j + 1];
            const double an = fabs(nu);
            L.dval[j] = an;
            L.dinv[g] = q_rcp(an);
            if (abort_get(L.s_abort) != 2 && C.real) g_put_sel(C.mb + 2 * (MB_D + g), -nu, C.tag, C.local);
        }
        if (last && E < 3) qrb_publish_v<W>(L, C, ODD, j);
    }
}

template <int W, int E, int ROLE>
__device__ __forceinline__ void qrb_era_role(double (&x)[32], double (&r)[4], const QbLds &L, const QbStepCtx &C)
{
    for (int jp = 0; jp < 4; ++jp) {
        qrb_step<W, E, ROLE, 0>(x, r, L, C, 8 * E + 2 * jp);
        qrb_step<W, E, ROLE, 1>(x, r, L, C, 8 * E + 2 * jp + 1);
    }
}
template <int W, int E>
__device__ __forceinline__ void qrb_era(double (&x)[32], double (&r)[4], const QbLds &L, const QbStepCtx &C, int wv)
{
    switch ((wv - E) & 3) {
    case 0: qrb_era_role<W, E, 0>(x, r, L, C); break;
    case 1: qrb_era_role<W, E, 1>(x, r, L, C); break;
    case 2: qrb_era_role<W, E, 2>(x, r, L, C); break;
    default: qrb_era_role<W, E, 3>(x, r, L, C); break;
    }
}

template <int W>
__device__ __forceinline__ void qrb_own(d4 (&c)[8], d4 (&z)[8], const QbLds &L, qword *mb, unsigned tag, bool local, int tid,
                                        double *__restrict__ Tout, int apply_pivot, int force_timeout, int part)
{
    constexpr int KB0 = 4 * W;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (scalar: the role switch is s_cmp, not exec masks)
    QRB_STAMP(48);
    // accumulator layout -> panel layout through the (free) panel buffer
    {
        const int nn = lane & 15, kq = lane >> 4, ct = wv & 1, rh = wv >> 1;
        double *cs = L.Vs + (16 * ct + nn) * VLD + kq;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[16 * (2 * u + rh) + 4 * r] = c[u][r];
    }
    if (tid < 32) {
#pragma unroll
        for (int k = 0; k < UST; ++k) L.Th[tid * UST + k] = 0.0;
    }
    QB_BARRIER();
    const int pc = tid >> 3, rg = tid & 7;
    double x[32];
    {
        const double2 *cs = reinterpret_cast<const double2 *>(L.Vs + pc * VLD + 2 * rg);
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            const double2 a = cs[4 * k];  // 16 (k >> 1) doubles further
            x[k] = a.x;
            x[k + 1] = a.y;
        }
    }
    // Z waits in the panel buffer while the panel is factored (64 registers less through the steps)
    QB_BARRIER();
    {
        double *zs = L.Vs + tid;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) zs[(4 * u + q) * 256] = z[u][q];
    }
    // Rows of the diagonal block that are finished leave x: r[kk] takes R[32 W + 16 (kk >> 1) + 2 rg + (kk & 1), my column]
    // and the register is zeroed, so that neither the norms nor the reflectors written to LDS need a mask.
    double r[4] = {0.0, 0.0, 0.0, 0.0};
    // the first column's raw reflector
    if (tid < 8) {
        double2 *dst = reinterpret_cast<double2 *>(L.ub + rg * UST);
#pragma unroll
        for (int k = KB0; k < 32; k += 2) dst[k >> 1] = make_double2(x[k], x[k + 1]);
    }
    QRB_STAMP(49);
    QbStepCtx C;
    C.mb = mb; C.tag = tag; C.local = local; C.real = true; C.lane = lane; C.pc = pc; C.rg = rg;
    C.stop_at = (force_timeout >= 2 && force_timeout < 1000 && part == 3) ? ((force_timeout - 2) & 31) : -1;
#ifdef QRB_X_EXTRA  // timing experiment (results are garbage): the 32 steps are run again without publishing
    for (int rep = 0; rep < 1 + (force_timeout >= 1000 ? force_timeout - 1000 : 0); ++rep) {
        C.real = rep == 0;
#endif
        qrb_era<W, 0>(x, r, L, C, wv);
        QRB_STAMP(50);
        qrb_era<W, 1>(x, r, L, C, wv);
        QRB_STAMP(51);
        qrb_era<W, 2>(x, r, L, C, wv);
        QRB_STAMP(52);
        qrb_era<W, 3>(x, r, L, C, wv);
#ifdef QRB_X_EXTRA
    }
#endif
    QB_BARRIER();
    QRB_STAMP(53);
    if (wv == 0) qrb_tcolumn<W>(L, mb, tag, local, lane, 31);  // (first: the next owner is waiting for this column of T)
    {
        const double *zs = L.Vs + tid;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) z[u][q] = zs[(4 * u + q) * 256];
    }
    // T = D^-1 R of my column: rows <= its position (UDT.jl:283-306), zeros below.  Rows above the panel are in x, the rows of
    // the diagonal block above the diagonal in r, the diagonal entry is -nu
    {
        const int gc = 32 * W + pc;  // position of my column
        const int dc = apply_pivot ? L.ord[gc] : gc;
        const double nu_c = L.rcps[32 + 2 * pc + 1];
        double2 *tcol = reinterpret_cast<double2 *>(Tout + (long)256 * dc + 2 * rg);
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            const int ra = 16 * (k >> 1) + 2 * rg;
            double a = 0.0, b = 0.0;
            if (k < KB0) {
                const double2 di = *reinterpret_cast<const double2 *>(L.dinv + ra);
                a = x[k] * di.x;
                b = x[k + 1] * di.y;
            } else if (k < KB0 + 4) {
                const double2 di = *reinterpret_cast<const double2 *>(L.dinv + ra);
                const double va = ra == gc ? -nu_c : r[k - KB0], vb = ra + 1 == gc ? -nu_c : r[k - KB0 + 1];
                a = ra <= gc ? va * di.x : 0.0;
                b = ra + 1 <= gc ? vb * di.y : 0.0;
            }
            tcol[4 * k] = make_double2(a, b);
        }
    }
    QRB_STAMP(54);
}

// the program of part W: panels in order, my own panel in between.  (W is a template constant so that the compiler sees
// what is live where: my columns C are dead once my panel is factored, and Z alone goes through the later panels.)
template <int W, int P>
__device__ __forceinline__ void qrb_panel(d4 (&c)[8], d4 (&z)[8], const QbLds &L, qword *mb, unsigned tag, bool local, int tid,
                                          double *__restrict__ Tout, int apply_pivot, int force_timeout)
{
    const int wv = tid >> 6, lane = tid & 63;
    if (P == W) {
        qrb_own<P>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout, W);
        QB_BARRIER();
        if (P > 0) {  // the update of Z that made way for my panel: its second half (the first went in while I waited, below)
            qrb_fetch<(P > 0 ? P - 1 : 0), 1, false>(mb, tag, L, tid);  // (published long ago: nothing to poll for)
            qrb_apply<(P > 0 ? P - 1 : 0), false, true, 1>(c, z, L, wv, lane);
        }
    }
    if (P >= W) {
        qrb_fetch<P, -1, (P != W)>(mb, tag, L, tid);  // (my own panel is there already)
        qrb_apply<P, false, true>(c, z, L, wv, lane);
    } else if (P == W - 1) {
        // I am the next owner: the first 16 reflectors are applied while the second 16 are still being factored
        qrb_fetch<P, 0>(mb, tag, L, tid);
        qrb_apply<P, true, false, 0>(c, z, L, wv, lane);
        // Z takes the same 16 reflectors now, in the time I would spend waiting for the other 16 (≈10 us of the owner's steps
        // against ≈5 us for both applications): half of the deferred update leaves the tail behind my own panel
        qrb_apply<P, false, true, 0>(c, z, L, wv, lane);
        qrb_fetch<P, 1, false>(mb, tag, L, tid);
        qrb_apply<P, true, false, 1>(c, z, L, wv, lane);
#ifdef QRB_X_DUMP
        if (W == 1) {
            const int nn = lane & 15, kq = lane >> 4, ct = wv & 1, rh = wv >> 1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) qrb_dump[(16 * (2 * u + rh) + 4 * r + kq) + 256 * (16 * ct + nn)] = c[u][r];
        }
#endif
    } else {
        qrb_fetch<P>(mb, tag, L, tid);
        qrb_apply<P, true, true>(c, z, L, wv, lane);
    }
}
template <int W>
__device__ __forceinline__ void qrb_part(d4 (&c)[8], d4 (&z)[8], const QbLds &L, qword *mb, unsigned tag, bool local, int tid,
                                         double *__restrict__ Tout, int apply_pivot, int force_timeout)
{
    qrb_panel<W, 0>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 1>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 2>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 3>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 4>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 5>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 6>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
    qrb_panel<W, 7>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout);
}

}  // namespace

__global__ __launch_bounds__(QB_THREADS) void qrb_udt_kernel(int n_units, const double *__restrict__ Aall, long strideA,
                                                            double *__restrict__ Uall, long strideU,
                                                            double *__restrict__ Dall, long strideD,
                                                            double *__restrict__ Tall, long strideT,
                                                            int *__restrict__ pivall, qword *mailbox, unsigned tag,
                                                            int *errflag, int apply_pivot, int force_sc1, int force_timeout,
                                                            const double *__restrict__ Ball, long strideB)
{
    extern __shared__ __attribute__((aligned(16))) double qb_lds[];
    QbLds L;
    {
        double *p = qb_lds;
        L.Vs = p; p += 32 * VLD;
        L.Ts = p; p += 32 * TLD;
        L.Wx = p; p += 4 * 16 * 64;
        L.ub = p; p += 2 * 8 * UST;
        L.scal = p; p += 4;
        L.dotbuf = p; p += 64;
        L.Th = p; p += 32 * UST;
        L.rcps = p; p += 96;
        L.dinv = p; p += 256;
        L.dval = p; p += 32;
        L.nrm = p; p += 256;
        int *q = reinterpret_cast<int *>(p);
        L.ord = q; q += 256;
        L.xcc = q; q += 8;
        L.s_abort = q; q += 4;
        L.fst = reinterpret_cast<long long *>(q);
    }
    const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
    const int unit = (seq / 8) * 8 + xcd, part = seq % 8;
    if (unit >= n_units) return;  // all parts of a missing unit leave together
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const double *__restrict__ A = Aall + (long)unit * strideA;
    qword *mb = mailbox + (long)unit * MB_GRANULES * 2;
    L.stamps = nullptr;
#ifdef QRB_STAMPS
    if (unit == 0 && qrb_stamp_ptr) L.stamps = qrb_stamp_ptr + part * 64;
#endif
    QRB_STAMP(0);
    if (tid == 0) abort_set(L.s_abort, 0);
    if (force_timeout == 1) {  // test hook: a launch whose hand-offs all time out at once
        if (tid == 0) atomicOr(errflag, 16);
        return;
    }
    __syncthreads();

    d4 c[8], z[8];
    // ---- column norms of the input: my 32 original columns, then everybody's through the mailbox ----
    {
        const int cc = tid >> 3, rg = tid & 7, col = 32 * part + cc;
        const double2 *ap = reinterpret_cast<const double2 *>(A + (long)col * 256) + rg;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double2 a = ap[8 * q];
            s0 = __builtin_fma(a.x, a.x, s0);
            s1 = __builtin_fma(a.y, a.y, s1);
        }
        const double nrm = qsum8(s0 + s1);
        if (rg == 0) g_put<false>(mb + 2 * (MB_N + col), nrm, tag);
    }
    if (tid == 0) {
        const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        g_put<false>(mb + 2 * (MB_X + part), (double)my_xcc, tag);
    }
    QRB_STAMP(1);
    L.nrm[tid] = g_wait(mb + 2 * (MB_N + tid), tag, L.s_abort);
    L.ord[tid] = tid;
    if (tid < 8) L.xcc[tid] = (int)g_wait(mb + 2 * (MB_X + tid), tag, L.s_abort);
    __syncthreads();
    QRB_STAMP(2);
    bool local = force_sc1 == 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) local = local && L.xcc[q] == L.xcc[0];
    // ---- position of every column: descending norm, first maximum first (UDT.jl:151-168 for the first step).  Every part
    // ranks its own 32 columns (8 lanes per column, 32 comparisons each) and the 256 positions cross in a second exchange:
    // 0.3 us + one hop instead of 256 comparisons per thread (7 us)
    {
        const int cc = tid >> 3, rg = tid & 7, col = 32 * part + cc;
        const double my = L.nrm[col];
        const double2 *np2 = reinterpret_cast<const double2 *>(L.nrm + 32 * rg);
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            const double2 o = np2[k >> 1];
            const int k0 = 32 * rg + k;
            cnt += (o.x > my) | ((o.x == my) & (k0 < col));
            cnt += (o.y > my) | ((o.y == my) & (k0 + 1 < col));
        }
        cnt += __builtin_amdgcn_mov_dpp(cnt, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        cnt += __builtin_amdgcn_mov_dpp(cnt, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
        cnt += __builtin_amdgcn_mov_dpp(cnt, 0x141, 0xf, 0xf, true);  // row_half_mirror
        if (rg == 0) g_put<false>(mb + 2 * (MB_P + col), (double)cnt, tag);
        const int rank = (int)g_wait(mb + 2 * (MB_P + tid), tag, L.s_abort);
        L.ord[rank & 255] = tid;  // (ranks are a permutation unless a norm is NaN; ord starts as the identity)
    }
    __syncthreads();
    QRB_STAMP(3);
    if (part == 0) pivall[(long)unit * 256 + tid] = L.ord[tid];

    // ---- my 32 columns (positions 32 part ..) in accumulator layout ----
    {
        const int nn = lane & 15, kq = lane >> 4, ct = wv & 1, rh = wv >> 1;
        const int pos = 32 * part + 16 * ct + nn;
        const double *ac = A + (long)L.ord[pos] * 256 + kq;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * u + rh) + 4 * r + kq;
                c[u][r] = ac[row - kq];
            }
    }
    // Z starts as my columns of the identity (U = Q), or of B' when the caller wants B Q (the product the reference forms right
    // behind the decomposition, stack.jl:360 / :378) - the block reflectors do not care.  Requested in one batch with my columns
    // of A just above
    {
        const int nn = lane & 15, kq = lane >> 4, ct = wv & 1, rh = wv >> 1;
        const int pos = 32 * part + 16 * ct + nn;
        if (Ball) {  // (one uniform branch around the 32 requests: a select per element made each a load -> wait of its own)
            const double *bp = Ball + (long)unit * strideB + pos + 256 * (16 * rh + kq);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[u][r] = bp[256 * (32 * u + 4 * r)];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[u][r] = 16 * (2 * u + rh) + 4 * r + kq == pos ? 1.0 : 0.0;
        }
    }
    double *__restrict__ Tout = Tall + (long)unit * strideT;
    QRB_STAMP(4);
    switch (part) {
    case 0: qrb_part<0>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 1: qrb_part<1>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 2: qrb_part<2>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 3: qrb_part<3>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 4: qrb_part<4>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 5: qrb_part<5>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    case 6: qrb_part<6>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    default: qrb_part<7>(c, z, L, mb, tag, local, tid, Tout, apply_pivot, force_timeout); break;
    }

    // ---- U = Q: Z holds columns 32 part .. of Q', i.e. rows 32 part .. of Q ----
    {
        const int nn = lane & 15, kq = lane >> 4, ct = wv & 1, rh = wv >> 1;
        double *uo = Uall + (long)unit * strideU + 32 * part + 16 * ct + nn;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) uo[(long)256 * (16 * (2 * u + rh) + 4 * r + kq)] = z[u][r];
    }
    if (tid < 32) Dall[(long)unit * strideD + 32 * part + tid] = L.dval[tid];
    QRB_STAMP(60);
    if (tid == 0 && abort_get(L.s_abort)) atomicOr(errflag, 16);
}
#ifdef QRB_FINE
extern "C" int dqmc_debug_qrb_fine(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qrb_fine_ptr), &devptr, sizeof(void *));
}
#endif
#ifdef QRB_STAMPS
extern "C" int dqmc_debug_qrb_stamps(void *devptr)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qrb_stamp_ptr), &devptr, sizeof(void *));
}
#endif

#if defined(QRB_X_DUMP) || defined(QRB_X_DUMP2)
extern "C" int dqmc_debug_qrb_dump(double *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(qrb_dump), sizeof(double) * 256 * 32);
}
#endif
size_t qrb_lds_bytes()
{
    return (size_t)(32 * VLD + 32 * TLD + 4 * 16 * 64 + 2 * 8 * UST + 4 + 64 + 32 * UST + 96 + 256 + 32 + 256) * sizeof(double) +
           (256 + 8 + 4) * sizeof(int)
#ifdef QRB_FINE
           + 4 * 32 * 8 * sizeof(long long)
#endif
        ;
}
size_t qrb_mailbox_bytes(int n_units) { return (size_t)((n_units + 7) / 8) * 8 * MB_GRANULES * 16; }

// workgroups of the kernel that one CU holds (its LDS admits one)
int qrb_blocks_per_cu()
{
    static int cached = -1;
    if (cached >= 0) return cached;
    (void)hipFuncSetAttribute((const void *)qrb_udt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qrb_lds_bytes());
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qrb_udt_kernel, QB_THREADS, qrb_lds_bytes()) != hipSuccess) nb = 0;
    cached = nb;
    return nb;
}

hipError_t launch_udt_blocked(int n_units, const double *A, long strideA, double *U, long strideU, double *D, long strideD,
                              double *T, long strideT, int *pivot, QrCoopWorkspace *ws, int apply_pivot, hipStream_t s,
                              const double *B, long strideB)
{
    if (!ws || !ws->mailbox2) return hipErrorInvalidValue;
    const int blocks = ((n_units + 7) / 8) * 64;
    if (blocks > ws->blk_max_blocks) return hipErrorInvalidValue;
    ws->epoch += 1;
    const unsigned tag = (unsigned)(ws->epoch & 0xffffffffull);
    hipLaunchKernelGGL(qrb_udt_kernel, dim3(blocks), dim3(QB_THREADS), qrb_lds_bytes(), s, n_units, A, strideA, U, strideU, D,
                       strideD, T, strideT, pivot, reinterpret_cast<qword *>(ws->mailbox2), tag, ws->errflag, apply_pivot,
                       ws->force_sc1, ws->force_timeout, B, strideB);
    return hipGetLastError();
}

}  // namespace dqmc
