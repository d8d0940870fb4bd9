// engine.cpp — host side of libdqmc_hip.so: device state of a batch of walkers, the
// propagate state machine of src/flavors/DQMC/stack.jl:502-631 expressed as batched
// kernel launches on one HIP stream, and the C ABI of include/dqmc_hip.h.
//
// All walkers of a handle move in lockstep (same current_slice / direction), so the
// control flow lives on the host and is identical to the reference's; the data of all
// walkers is processed by every launch (grid = units x tiles).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dqmc_hip.h"
#include "kernels.h"
#include <rccl/rccl.h>

using namespace dqmc;

struct dqmc_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
};

static thread_local std::string g_create_error;

struct UTStack;  // unequal-time stack (unequal_time.inl)

struct dqmc_handle {
    dqmc_params p{};
    int n = 0, nb = 1, N = 0, M = 0, s = 0, K = 0, W = 0, units = 0, kd = 32;
    long nn = 0;
    double lambda = 0, epl = 0, eml = 0;
    SweepConsts sc{};
    hipStream_t stream = nullptr;
    // constants (nb x n x n)
    double *eT = nullptr, *eTinv = nullptr, *eT2 = nullptr, *eTinv2 = nullptr;
    double *eT2T = nullptr;  // transposed copy of eT2: A operand of the daggered slice products in slab.hip
    bool slab = false;       // n == 256 dense path: slice chains and wraps as slab-resident launches
    int8_t *conf = nullptr;  // W x (N x M)
    // stack (slot-major): u/t: (K+1) x units x n^2 ; d: (K+1) x units x n
    double *u_stack = nullptr, *t_stack = nullptr, *d_stack = nullptr;
    // slot i of the UDT stack = (su[i], sd[i], st[i]); entry K + 1 is a spare: a slot is rewritten by building the new
    // factors in the spare and swapping the pointers, so the old content stays readable (no load_slot copies)
    std::vector<double *> su, sd, st;
    double *Ul = nullptr, *Ur = nullptr, *Tl = nullptr, *Tr = nullptr, *greens = nullptr, *greens_temp = nullptr;
    double *tmp1 = nullptr, *tmp2 = nullptr, *bufA = nullptr, *bufB = nullptr;
    // scratch of one UDT (udt_AVX_pivot!): V (Householder vectors; hand-over buffer of the two-phase QR before that),
    // W (factored matrix of the cooperative QR, then the compact-WY product), S (V'V), tau, pivot, and what the
    // triangular solves need (winv: inverted 16 x 16 diagonal blocks; ts: n > 256, panel-solved copy of the right-hand
    // side).
    struct QrSet {
        double *V = nullptr, *W = nullptr, *S = nullptr, *tau = nullptr, *winv = nullptr, *ts = nullptr;
        int *pivot = nullptr;
    } qs[1];
    double *&qrV = qs[0].V, *&qrW = qs[0].W, *&qrS = qs[0].S, *&trsm_w = qs[0].winv, *&trsm_s = qs[0].ts, *&tau = qs[0].tau;
    int *&pivot = qs[0].pivot;
    double *Dl = nullptr, *Dr = nullptr;
    hipStream_t cur = nullptr;  // the stream the launch helpers use (= stream)
    double *sU = nullptr, *sVT = nullptr;
    double *greens_alt = nullptr, *lu_img = nullptr;  // decide / apply sweep (sweep_lu.hip)
    bool sweep_lu = true, sweep_fused = true;
    WalkerRng *rng = nullptr;
    DevStats *stats = nullptr;
    unsigned long long *pc_scratch = nullptr;  // prop_check_kernel: partial maximum + arrival counter per walker
    std::vector<double *> uniforms;  // per walker device arrays
    double *acc = nullptr;
    size_t acc_n = 0;
    // correlation measurements (EachSitePairByDistance tables set by the host)
    int n_dirs = 0;
    int *dir_ptr = nullptr, *pair_src = nullptr, *pair_trg = nullptr;
    double *corr_per_walker = nullptr, *corr_acc = nullptr;
    int K_loc = 0;                  // EachLocalQuadByDistance{K}
    int *trg_of = nullptr;          // [K][n]
    size_t pc_n = 0;
    double *pc_per_walker = nullptr, *pc_acc = nullptr;
    size_t corr_n = 0;
    int current_slice = 0, direction = 0;
    bool prepared = false;
    long long conf_version = 0;     // bumped whenever the HS field changes (mc.last_sweep's role for the UT stack)
    UTStack *ut = nullptr;
    std::string err;
    // timing
    bool timing = false;
    struct Ev { hipEvent_t a, b; int fam; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    double fam_ms[DQMC_K_COUNT] = {0};
    long long fam_n[DQMC_K_COUNT] = {0};
    std::vector<void *> allocs;
    QrCoopWorkspace qr_ws;
    int qrb_sites = 0;  // call sites of udt_AVX_pivot! that take the one-launch blocked form (DQMC_QRB_SITES)
    // checkerboard products with sparse bond-group factors (cb.hip); the dense constants above stay valid
    struct {
        bool on = false;
        int kmax = 0, n_mats = 0;
        double *vals = nullptr, *mu = nullptr, *mu_inv = nullptr;
        int *cols = nullptr;
        int seq[7][32];
        int len[7] = {0, 0, 0, 0, 0, 0, 0};
    } cb;
    // measurement reduction (dqmc_reduce): packed device buffer [sums | maxima | minima]
    double *red_buf = nullptr;
    size_t red_cap = 0;
    dqmc_stats red_stats{};
    bool red_valid = false;
    size_t red_sizes[4] = {0, 0, 0, 0};  // section sizes of the LAST reduction (dqmc_get_reduced checks against these)
};

// ---------------------------------------------------------------------------
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            h->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                \
            return DQMC_ERR_HIP;                                                                       \
        }                                                                                              \
    } while (0)
#define CHK(expr)                \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != 0) return rc_; \
    } while (0)

static int fail(dqmc_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

template <typename T>
static int dalloc(dqmc_handle *h, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    h->allocs.push_back(q);
    if (zero) HIPCHK(hipMemsetAsync(q, 0, (count ? count : 1) * sizeof(T), h->stream));
    *p = (T *)q;
    return 0;
}

// ---- timing scopes ---------------------------------------------------------
static int timing_drain(dqmc_handle *h)
{
    if (h->pending.empty()) return 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto &e : h->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        h->fam_ms[e.fam] += ms;
        h->fam_n[e.fam] += 1;
        h->pool.push_back(e.a);
        h->pool.push_back(e.b);
    }
    h->pending.clear();
    return 0;
}
struct Timed {
    dqmc_handle *h;
    int fam;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(dqmc_handle *h_, int fam_) : h(h_), fam(fam_)
    {
        if (!h->timing) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, h->cur);
    }
    ~Timed()
    {
        if (!h->timing) return;
        (void)hipEventRecord(b, h->cur);
        h->pending.push_back({a, b, fam});
        if (h->pending.size() >= 2048) (void)timing_drain(h);
    }
};

// ---- argument builders -------------------------------------------------------
static GemmArgs gemm_base(dqmc_handle *h, MatRef A, int tA, MatRef B, int tB, double *C)
{
    GemmArgs g{};
    g.M = g.N = g.K = h->n;
    g.n_units = h->units;
    g.nb = h->nb;
    g.A = A;
    g.B = B;
    g.C = C;
    g.strideC = h->nn;
    g.ldc = h->n;
    g.transA = tA;
    g.transB = tB;
    g.kscale = g.colscale = g.rowscale = g.adddiag = vs_none();
    g.row_first = 0;
    g.alpha = 1.0;
    g.ident = 0.0;
    g.beta = 0;
    g.tri = 0;
    return g;
}
static MatRef U_(dqmc_handle *h, const double *p) { return mat(p, h->nn, h->n); }        // per unit
static MatRef C_(dqmc_handle *h, const double *p) { return mat(p, 0, h->n, h->nn); }     // shared constant, per block
// a pair of events for a dispatch-attached (kernel-only) timing, or nulls when timing is off
static void timing_events(dqmc_handle *h, hipEvent_t *a, hipEvent_t *b)
{
    *a = *b = nullptr;
    if (!h->timing) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    *a = get();
    *b = get();
}
static int timing_push(dqmc_handle *h, hipEvent_t a, hipEvent_t b, int fam)
{
    if (!a) return 0;
    h->pending.push_back({a, b, fam});
    if (h->pending.size() >= 2048) return timing_drain(h);
    return 0;
}
static int run_gemm(dqmc_handle *h, const GemmArgs &g)
{
    if (!h->timing) {
        HIPCHK(launch_gemm(g, h->cur));
        return 0;
    }
    // kernel-only duration: the events are attached to the dispatch itself (no launch gap inside)
    auto get = [&]() {
        hipEvent_t e;
        if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    hipEvent_t a = get(), b = get();
    HIPCHK(launch_gemm(g, h->cur, a, b));
    h->pending.push_back({a, b, DQMC_K_GEMM});
    if (h->pending.size() >= 2048) CHK(timing_drain(h));
    return 0;
}
// exp(sign*lambda*conf[:,slice]) for block 0, exp(-sign*lambda*conf) for block 1
// (HubbardModelAttractive.jl:100-110, HubbardModelRepulsive.jl:113-126); slice 1-based
static VecSrc vs_conf(dqmc_handle *h, int slice, int sign)
{
    VecSrc v{};
    v.mode = 2;
    v.conf = h->conf + (long)(slice - 1) * h->N;
    v.conf_stride = (long)h->N * h->M;
    const double a = sign > 0 ? h->epl : h->eml, b = sign > 0 ? h->eml : h->epl;
    v.cpos[0] = a; v.cneg[0] = b;
    v.cpos[1] = b; v.cneg[1] = a;
    return v;
}
static double *uslot(dqmc_handle *h, int i) { return h->su[i]; }
static double *tslot(dqmc_handle *h, int i) { return h->st[i]; }
static double *dslot(dqmc_handle *h, int i) { return h->sd[i]; }
struct Udt { const double *u, *d, *t; };
static Udt slot_ref(dqmc_handle *h, int i) { return Udt{h->su[i], h->sd[i], h->st[i]}; }
static void slot_swap_spare(dqmc_handle *h, int i)
{
    const int sp = h->K + 1;
    std::swap(h->su[i], h->su[sp]); std::swap(h->sd[i], h->sd[sp]); std::swap(h->st[i], h->st[sp]);
}

static int copy_mat(dqmc_handle *h, double *dst, const double *src)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * h->units * h->nn, hipMemcpyDeviceToDevice, h->cur));
    return 0;
}
static int copy_vec(dqmc_handle *h, double *dst, const double *src)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * h->units * h->n, hipMemcpyDeviceToDevice, h->cur));
    return 0;
}
static int set_identity(dqmc_handle *h, double *A)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(launch_set_identity(h->n, h->units, A, h->nn, h->cur));
    return 0;
}
static int set_ones(dqmc_handle *h, double *d)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(launch_fill(d, (size_t)h->units * h->n, 1.0, h->cur));
    return 0;
}

static int alloc_qr_workspace(dqmc_handle *h)
{
    // device-side error word (bit 1: a hand-off inside the sweep elimination kernel timed out; bit 0 is no longer set by
    // anything: a cooperative-QR time-out is not an error since round 2, the guarded kernel behind the launch redoes
    // the factorisation and dqmc_qr_fallbacks counts it)
    CHK(dalloc(h, &h->qr_ws.errflag, (size_t)1));
    if (h->n > 256) return 0;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->p.device_id));
    const size_t slots = (size_t)((h->units + 7) / 8) * 8 * QR_COOP_SLOTS_PER_UNIT;
    CHK(dalloc(h, &h->qr_ws.mailbox, slots * QR_COOP_SLOT));
    CHK(dalloc(h, &h->qr_ws.fb, (size_t)2));
    // co-residency: what the occupancy API reports for the kernel on this device (its ~200 VGPRs admit 2 per CU)
    h->qr_ws.max_blocks = prop.multiProcessorCount * qr_coop_blocks_per_cu();
    h->qr_ws.epoch = 0;
    if (const char *e = getenv("DQMC_QR_TAIL")) h->qr_ws.tail_j0 = atoi(e);  // A/B switches, read per handle
    h->qr_ws.force_sc1 = getenv("DQMC_QR_SC1") != nullptr;
    h->qr_ws.no_coop = getenv("DQMC_QR_NOCOOP") != nullptr;
    if (const char *e = getenv("DQMC_QR_FORCE_TIMEOUT"))
        h->qr_ws.force_timeout = strncmp(e, "step:", 5) == 0 ? 2 + atoi(e + 5) : (strncmp(e, "extra:", 6) == 0 ? 1000 + atoi(e + 6) : 1);
    // pre-pivoted blocked UDT in one launch (qrb.hip): n == 256, all eight workgroups of every unit co-resident.
    // DQMC_QR_NOBLOCKED: off; DQMC_QRB_SITES: bit mask of the call sites that use it (1 = slice-sequence builds and every other
    // caller, 2 = first, 4 = second factorisation of calculate_greens_AVX!)
    h->qrb_sites = 0;
    if (h->n == 256 && getenv("DQMC_QR_NOBLOCKED") == nullptr) {
        const int per_cu = qrb_blocks_per_cu();
        if (per_cu >= 1 && ((h->units + 7) / 8) * 64 <= prop.multiProcessorCount * per_cu) {
            CHK(dalloc(h, (char **)&h->qr_ws.mailbox2, qrb_mailbox_bytes(h->units)));
            h->qr_ws.blk_max_blocks = prop.multiProcessorCount * per_cu;
            h->qrb_sites = 7;
            if (const char *e = getenv("DQMC_QRB_SITES")) h->qrb_sites = atoi(e) & 7;
        }
    }
    return 0;
}
static int check_qr_workspace(dqmc_handle *h)
{
    if (!h->qr_ws.errflag) return 0;
    int e = 0;
    HIPCHK(hipMemcpy(&e, h->qr_ws.errflag, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {  // reported once: the flag is cleared, the data of the failed call is not trustworthy
        HIPCHK(hipMemset(h->qr_ws.errflag, 0, sizeof(int)));
        if (e & 16)
            return fail(h, DQMC_ERR_HIP, "blocked UDT: a hand-off between the workgroups of a matrix timed out (results of this call "
                                         "are invalid; DQMC_QR_NOBLOCKED=1 selects the kernels without that requirement)");
        return fail(h, DQMC_ERR_HIP, (e & 8) ? "site sweep (one launch per slice): a hand-over between workgroups timed out - the grid was "
                                               "not co-resident (results of this call are invalid; the default launch-per-chunk form has no "
                                               "such requirement)"
                                             : "sweep elimination: hand-off timed out (results of this call are invalid)");
    }
    return 0;
}

// ---- UDT (udt_AVX_pivot!, src/linalg/UDT.jl:192-306) ---------------------------
// A is factored (in place, or into q.W by the cooperative kernel).  Q is formed in compact-WY form with GEMMs instead
// of the reference's reflector-by-reflector back accumulation (UDT.jl:250-266):
//   Q = H_1...H_n = I - V S^-1 V',  S = striu(V'V) + diag(1/tau)
// udt_factor: "QR decomposition" loop + D, T (and V for the second half); udt_formq: U = Q from V, tau of the set.
typedef dqmc_handle::QrSet QrSet;
static int udt_factor(dqmc_handle *h, double *A, double *Dout, double *Tout, int apply, QrSet &q)
{
    const int n = h->n;
    const double *F = A;  // where the factored matrix ends up (q.W behind the cooperative QR)
    {
        Timed t(h, DQMC_K_QR);
        // q.V is free until udt_finish writes V: it serves as the hand-over buffer of the two-phase QR
        HIPCHK(launch_qr_pivot(n, h->units, A, h->nn, q.tau, q.pivot, &h->qr_ws, q.W, h->nn, &F, h->cur, q.V, h->nn));
    }
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_udt_finish(n, h->units, A, h->nn, F, h->nn, q.pivot, Dout, n, q.V, h->nn, Tout, h->nn, apply,
                                 h->cur));
    }
    return 0;
}
static int udt_formq(dqmc_handle *h, double *Uout, QrSet &q, double *winv, double *ts)
{
    const int n = h->n;
    GemmArgs g = gemm_base(h, U_(h, q.V), 1, U_(h, q.V), 0, q.S);
    g.tri = (n % 64 == 0) ? 1 : 0;  // V is unit lower triangular: a third of the k range on average
    CHK(run_gemm(h, g));
    {
        Timed t(h, DQMC_K_TRSM);
        HIPCHK(launch_trsm_right_upper(n, h->units, q.V, h->nn, q.S, h->nn, nullptr, q.tau, n, q.W, h->nn, winv, h->cur, ts));
    }
    g = gemm_base(h, U_(h, q.W), 0, U_(h, q.V), 1, Uout);
    g.alpha = -1.0;
    g.ident = 1.0;
    g.tri = (n % 64 == 0) ? 2 : 0;
    CHK(run_gemm(h, g));
    return 0;
}
// n == 256: the whole of udt_AVX_pivot! in one launch, pivot order fixed up front (qrb.hip).  site: see DQMC_QRB_SITES
static bool udt_is_fused(const dqmc_handle *h, int site) { return (h->qrb_sites >> site) & 1; }
// B != null: Uout = B Q (Uout must not be B)
static int udt_fused(dqmc_handle *h, const double *A, double *Uout, double *Dout, double *Tout, int apply, QrSet &q,
                     const double *B = nullptr)
{
    Timed t(h, DQMC_K_QR);
    HIPCHK(launch_udt_blocked(h->units, A, h->nn, Uout, h->nn, Dout, h->n, Tout, h->nn, q.pivot, &h->qr_ws, apply, h->cur, B,
                              h->nn));
    return 0;
}
static int udt(dqmc_handle *h, double *A, double *Uout, double *Dout, double *Tout, int apply)
{
    if (udt_is_fused(h, 0) && Tout && Tout != A) return udt_fused(h, A, Uout, Dout, Tout, apply, h->qs[0]);
    CHK(udt_factor(h, A, Dout, Tout, apply, h->qs[0]));
    return udt_formq(h, Uout, h->qs[0], h->qs[0].winv, h->qs[0].ts);
}
// Out = A[:, pivot of set q] / triu(T) (rdivp!, general.jl:138-166)
static int rdivp_set(dqmc_handle *h, const double *A, const double *T, double *Out, const QrSet &q, double *winv, double *ts)
{
    Timed t(h, DQMC_K_TRSM);
    HIPCHK(launch_trsm_right_upper(h->n, h->units, A, h->nn, T, h->nn, q.pivot, nullptr, 0, Out, h->nn, winv, h->cur, ts));
    return 0;
}
static int rdivp(dqmc_handle *h, double *A, const double *T)
{
    return rdivp_set(h, A, T, A, h->qs[0], h->qs[0].winv, h->qs[0].ts);
}

// ---- calculate_greens_AVX! (stack.jl:337-393) -----------------------------------
// L = (Ul, Dl, Tl), R = (Ur, Dr, Tr) are only read (they may be stack slots); h->Ul .. h->Tr are the work matrices
// the reference overwrites its six inputs with.  L / R may also BE those work matrices (each is consumed before the
// step that overwrites it).
static int calculate_greens_src(dqmc_handle *h, double *out, Udt L, Udt R)
{
    const int n = h->n;
    QrSet &qa = h->qs[0], &qb = h->qs[0];
    GemmArgs g = gemm_base(h, U_(h, L.t), 0, U_(h, R.t), 1, out);  // :346-348
    g.colscale = vs_arr(R.d, n);
    g.rowscale = vs_arr(L.d, n);
    CHK(run_gemm(h, g));
    if (udt_is_fused(h, 1)) {
        // :349 and :360 in one launch: T out of place (qa.W), and the kernel carries Ul' through the reflectors instead of the
        // identity, so that its "U" is Tl = Ul Q already (Q itself is not used again: :362 overwrites Tr)
        if (L.u == h->Tl) return fail(h, DQMC_ERR_STATE, "calculate_greens: Ul aliases Tl");
        CHK(udt_fused(h, out, h->Tl, h->Dr, qa.W, 0, qa, L.u));
        CHK(rdivp_set(h, R.u, qa.W, h->Ur, qa, qa.winv, qa.ts));       // :361
    } else {
        CHK(udt_factor(h, out, h->Dr, nullptr, 0, qa));                    // :349, first half
        CHK(rdivp_set(h, R.u, out, h->Ur, qa, qa.winv, qa.ts));            // :361 (out of place: Ur = R.u[:, p] / T)
        CHK(udt_formq(h, h->Tr, qa, qa.winv, qa.ts));                      // :349, second half
        CHK(run_gemm(h, gemm_base(h, U_(h, L.u), 0, U_(h, h->Tr), 0, h->Tl)));    // :360
    }
    g = gemm_base(h, U_(h, h->Tl), 1, U_(h, h->Ur), 0, h->Tr);         // :362 + :368
    g.adddiag = vs_arr(h->Dr, n);
    CHK(run_gemm(h, g));
    const double *tlul = h->Tr;  // where Tl Ul of :378 ends up
    if (udt_is_fused(h, 2)) {
        // :376 and :378 in one launch: "U" = Tl Q goes to Ul (free: the reference's Ul = Q is only used in :378)
        CHK(udt_fused(h, h->Tr, h->Ul, h->Dr, qb.W, 0, qb, h->Tl));
        CHK(rdivp_set(h, h->Ur, qb.W, h->Ur, qb, qb.winv, qb.ts));     // :377
        tlul = h->Ul;
    } else {
        CHK(udt_factor(h, h->Tr, h->Dr, nullptr, 0, qb));                  // :376
        CHK(rdivp_set(h, h->Ur, h->Tr, h->Ur, qb, qb.winv, qb.ts));        // :377
        CHK(udt_formq(h, h->Ul, qb, qb.winv, qb.ts));
        CHK(run_gemm(h, gemm_base(h, U_(h, h->Tl), 0, U_(h, h->Ul), 0, h->Tr)));  // :378
    }
    g = gemm_base(h, U_(h, h->Ur), 0, U_(h, tlul), 1, out);            // :382-391
    g.kscale = vs_inv(h->Dr, n);
    CHK(run_gemm(h, g));
    return 0;
}
static int calculate_greens(dqmc_handle *h, double *out)
{
    return calculate_greens_src(h, out, Udt{h->Ul, h->Dl, h->Tl}, Udt{h->Ur, h->Dr, h->Tr});
}

// ---- checkerboard products with sparse factors (slice_matrices.jl:104-222, DQMC.jl:731-750) --------------------
enum { CB_LEFT_B = 0, CB_LEFT_BINV = 1, CB_LEFT_BDAG = 2, CB_RIGHT_B = 3, CB_RIGHT_BINV = 4, CB_RIGHT_ET = 5,
       CB_LEFT_ETINV = 6 };
static int cb_mult(dqmc_handle *h, int which, int slice, const double *X, double *O, const double *qscale)
{
    CbArgs a{};
    a.n = h->n; a.nb = h->nb; a.kmax = h->cb.kmax;
    a.side = (which == CB_RIGHT_B || which == CB_RIGHT_BINV || which == CB_RIGHT_ET) ? 1 : 0;
    a.seq_len = h->cb.len[which];
    for (int i = 0; i < a.seq_len; ++i) a.seq[i] = h->cb.seq[which][i];
    a.vals = h->cb.vals; a.cols = h->cb.cols;
    a.X = X; a.O = O; a.strideX = h->nn;
    a.conf = slice >= 1 ? h->conf + (long)(slice - 1) * h->N : nullptr;
    a.conf_stride = (long)h->N * h->M;
    a.epl = h->epl; a.eml = h->eml;
    switch (which) {
    case CB_LEFT_B: a.pre_conf = +1; a.pre_vec = h->cb.mu; break;
    case CB_LEFT_BINV: a.post_conf = -1; a.post_vec = h->cb.mu_inv; break;
    case CB_LEFT_BDAG: a.post_conf = +1; a.post_vec = h->cb.mu; break;
    case CB_RIGHT_B: a.post_conf = +1; a.post_vec = h->cb.mu; break;
    case CB_RIGHT_BINV: a.pre_conf = -1; a.pre_vec = h->cb.mu_inv; break;
    default: break;
    }
    a.qscale = qscale; a.qstride = h->n;
    hipEvent_t ea, eb;
    timing_events(h, &ea, &eb);
    HIPCHK(launch_cb_apply(a, h->units, h->cur, ea, eb));
    return timing_push(h, ea, eb, DQMC_K_GEMM);
}

// ---- slab-resident product chains (slab.hip) -----------------------------------------
static SlabArgs slab_base(dqmc_handle *h, const double *X0, long x_su, long x_sb, double *out)
{
    SlabArgs a{};
    a.n_units = h->units; a.nb = h->nb; a.nsteps = 0;
    a.X0 = X0; a.x_su = x_su; a.x_sb = x_sb;
    a.out = out; a.out_su = h->nn;
    a.conf_stride = (long)h->N * h->M;
    a.epl = h->epl; a.eml = h->eml;
    return a;
}
static void slab_step_const(dqmc_handle *h, SlabArgs &a, const double *C)  // shared constant, per block
{
    SlabStep &st = a.st[a.nsteps++];
    st = SlabStep{};
    st.A = C; st.su = 0; st.sb = h->nn;
}
static void slab_step_unit(dqmc_handle *h, SlabArgs &a, const double *A)
{
    SlabStep &st = a.st[a.nsteps++];
    st = SlabStep{};
    st.A = A; st.su = h->nn; st.sb = 0;
}
static int run_slab(dqmc_handle *h, const SlabArgs &a)
{
    hipEvent_t ea, eb;
    timing_events(h, &ea, &eb);
    HIPCHK(launch_slab_chain(a, h->cur, ea, eb));
    return timing_push(h, ea, eb, DQMC_K_GEMM);
}

// ---- slice sequences (stack.jl:272-311, slice_matrices.jl:42-76) -------------------
// The s products of a stack interval only need the HS field of slices that sweep_spatial has already left behind, in
// the order the sweep visits them (up pass: B_l X for l = (idx-1)s+1 .. idx s; down pass: B_l' X for l = idx s ..
// (idx-1)s+1).
static const int8_t *conf_slice(dqmc_handle *h, int slice) { return h->conf + (long)(slice - 1) * h->N; }
static void chain_steps(dqmc_handle *h, SlabArgs &a, int dir, int idx, int t0, int t1)  // products t0 .. t1 - 1
{
    for (int t = t0; t < t1; ++t) {
        SlabStep &st = a.st[a.nsteps];
        if (dir == 1) {
            slab_step_const(h, a, h->eT2);
            st.pre_conf = conf_slice(h, (idx - 1) * h->s + 1 + t);
            st.pre_sign = +1;
        } else {  // B_l' X = eV (eT2' X)
            slab_step_const(h, a, h->eT2T);
            st.post_conf = conf_slice(h, idx * h->s - t);
            st.post_sign = +1;
        }
    }
}
static int wrap_greens_slab(dqmc_handle *h, const double *src, double *dst, int curr_slice, int direction);
// dir = +1: add_slice_sequence_left(idx), reads slot idx - 1, writes slot idx; dir = -1: add_slice_sequence_right(idx),
// reads slot idx, writes slot idx - 1 (idx 1-based as in the reference).  wrap_temp: the up pass's wrap of the old
// Green's function for the propagation check (stack.jl:534-536).
static int add_slice_sequence(dqmc_handle *h, int dir, int idx, bool wrap_temp)
{
    const int src = dir == 1 ? idx - 1 : idx, dst = dir == 1 ? idx : idx - 1;
    const int t0 = 0;
    const double *X = uslot(h, src);
    if (wrap_temp) {
        if (h->slab && !h->cb.on) CHK(wrap_greens_slab(h, h->greens, h->greens_temp, h->current_slice - 1, 1));
    }
    double *out = nullptr;
    if (h->slab && !h->cb.on && h->s <= SLAB_MAX_STEPS) {  // the (remaining) products in one launch
        out = h->bufA;
        SlabArgs a = slab_base(h, X, h->nn, 0, out);
        chain_steps(h, a, dir, idx, t0, h->s);
        a.col_d = dslot(h, src); a.col_stride = h->n;  // stack.jl:281 / :305
        CHK(run_slab(h, a));
    }
    else for (int t = 0; t < h->s; ++t) {
        const int slice = dir == 1 ? (idx - 1) * h->s + 1 + t : idx * h->s - t;
        out = (t & 1) ? h->bufB : h->bufA;
        if (h->cb.on) {
            CHK(cb_mult(h, dir == 1 ? CB_LEFT_B : CB_LEFT_BDAG, slice, X, out, t == h->s - 1 ? dslot(h, src) : nullptr));
            X = out;
            continue;
        }
        GemmArgs g = gemm_base(h, C_(h, h->eT2), dir == 1 ? 0 : 1, U_(h, X), 0, out);  // (eT2*eV)' = eV*eT2'
        if (dir == 1) g.kscale = vs_conf(h, slice, +1);
        else { g.rowscale = vs_conf(h, slice, +1); g.row_first = 1; }
        if (t == h->s - 1) g.colscale = vs_arr(dslot(h, src), h->n);
        CHK(run_gemm(h, g));
        X = out;
    }
    // new factors into the spare, then slot dst <-> spare: the old slot dst stays readable
    const int sp = h->K + 1;
    if (udt_is_fused(h, 0)) {
        CHK(udt_fused(h, out, uslot(h, sp), dslot(h, sp), h->tmp2, 1, h->qs[0]));
    } else {
        CHK(udt_factor(h, out, dslot(h, sp), h->tmp2, 1, h->qs[0]));
        CHK(udt_formq(h, uslot(h, sp), h->qs[0], h->qs[0].winv, h->qs[0].ts));
    }
    CHK(run_gemm(h, gemm_base(h, U_(h, h->tmp2), 0, U_(h, tslot(h, src)), 0, tslot(h, sp))));
    slot_swap_spare(h, dst);
    return 0;
}
static int add_slice_sequence_left(dqmc_handle *h, int idx, bool wrap_temp = false) { return add_slice_sequence(h, +1, idx, wrap_temp); }
static int add_slice_sequence_right(dqmc_handle *h, int idx) { return add_slice_sequence(h, -1, idx, false); }

// wrap_greens! (stack.jl:491-500), out of place, one launch: column slab c of the result is
//   +1:  eT2 (eV (G (eV^-1 eTinv2[:, c])))          -1:  eV^-1 (eTinv2 (G eT2[:, c])) eV[c]
static int wrap_greens_slab(dqmc_handle *h, const double *src, double *dst, int curr_slice, int direction)
{
    if (direction == -1) {
        const int8_t *c = conf_slice(h, curr_slice - 1);
        SlabArgs a = slab_base(h, h->eT2, 0, h->nn, dst);
        slab_step_unit(h, a, src);
        slab_step_const(h, a, h->eTinv2);
        a.st[1].post_conf = c; a.st[1].post_sign = -1;
        a.col_conf = c; a.col_sign = +1;
        return run_slab(h, a);
    }
    const int8_t *c = conf_slice(h, curr_slice);
    SlabArgs a = slab_base(h, h->eTinv2, 0, h->nn, dst);
    slab_step_unit(h, a, src);
    a.st[0].pre_conf = c; a.st[0].pre_sign = -1;
    slab_step_const(h, a, h->eT2);
    a.st[1].pre_conf = c; a.st[1].pre_sign = +1;
    return run_slab(h, a);
}
static int wrap_greens_inplace(dqmc_handle *h, double *gf, int curr_slice, int direction);
// *gf <- wrapped *gf (the slab path writes into tmp1 and exchanges the two pointers)
static int wrap_greens(dqmc_handle *h, double **gf, int curr_slice, int direction)
{
    if (h->slab && !h->cb.on) {
        CHK(wrap_greens_slab(h, *gf, h->tmp1, curr_slice, direction));
        std::swap(*gf, h->tmp1);
        return 0;
    }
    return wrap_greens_inplace(h, *gf, curr_slice, direction);
}
static int wrap_greens_inplace(dqmc_handle *h, double *gf, int curr_slice, int direction)
{
    if (h->cb.on) {  // both products in place, slab by slab
        const int l = direction == -1 ? curr_slice - 1 : curr_slice;
        CHK(cb_mult(h, direction == -1 ? CB_LEFT_BINV : CB_LEFT_B, l, gf, gf, nullptr));
        CHK(cb_mult(h, direction == -1 ? CB_RIGHT_B : CB_RIGHT_BINV, l, gf, gf, nullptr));
        return 0;
    }
    if (direction == -1) {
        const int l = curr_slice - 1;
        GemmArgs g = gemm_base(h, C_(h, h->eTinv2), 0, U_(h, gf), 0, h->tmp1);  // (eV^-1 eTinv2) * G
        g.rowscale = vs_conf(h, l, -1);
        CHK(run_gemm(h, g));
        g = gemm_base(h, U_(h, h->tmp1), 0, C_(h, h->eT2), 0, gf);              // . * (eT2 eV)
        g.colscale = vs_conf(h, l, +1);
        CHK(run_gemm(h, g));
    } else {
        const int l = curr_slice;
        GemmArgs g = gemm_base(h, C_(h, h->eT2), 0, U_(h, gf), 0, h->tmp1);     // (eT2 eV) * G
        g.kscale = vs_conf(h, l, +1);
        CHK(run_gemm(h, g));
        g = gemm_base(h, U_(h, h->tmp1), 0, C_(h, h->eTinv2), 0, gf);           // . * (eV^-1 eTinv2)
        g.kscale = vs_conf(h, l, -1);
        CHK(run_gemm(h, g));
    }
    return 0;
}

// identity factors in slot `slot`; the old content stays readable in the spare
static int reset_slot(dqmc_handle *h, int slot)
{
    const int sp = h->K + 1;
    CHK(set_identity(h, uslot(h, sp)));
    CHK(set_ones(h, dslot(h, sp)));
    CHK(set_identity(h, tslot(h, sp)));
    slot_swap_spare(h, slot);
    return 0;
}
static int prop_check(dqmc_handle *h)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(launch_prop_check(h->n, h->nb, h->W, h->greens_temp, h->greens, h->nn, h->stats, h->pc_scratch, h->cur));
    return 0;
}

// stack.jl:108-159 (the parts with observable effect)
static int init_stack(dqmc_handle *h)
{
    CHK(set_identity(h, h->Ul)); CHK(set_identity(h, h->Ur));
    CHK(set_identity(h, h->Tl)); CHK(set_identity(h, h->Tr));
    CHK(set_ones(h, h->Dl)); CHK(set_ones(h, h->Dr));
    h->current_slice = 0;
    h->direction = 0;
    return 0;
}
// stack.jl:242-255
static int build_stack(dqmc_handle *h)
{
    CHK(reset_slot(h, 0));
    for (int i = 1; i <= h->K; ++i) CHK(add_slice_sequence_left(h, i));
    h->current_slice = h->M + 1;
    h->direction = -1;
    return 0;
}

// stack.jl:502-631
static int propagate(dqmc_handle *h)
{
    const int M = h->M, s = h->s;
    if (h->direction == 1) {
        if (h->current_slice % s == 0) {
            h->current_slice += 1;
            if (h->current_slice == 1) {
                const Udt R = slot_ref(h, 0);   // copyto!(s.Ur, s.u_stack[1]) ... (stack.jl:512-520) without the copies
                CHK(reset_slot(h, 0));
                CHK(calculate_greens_src(h, h->greens, slot_ref(h, 0), R));
            } else if (1 < h->current_slice && h->current_slice <= M) {
                const int idx = (h->current_slice - 1) / s;
                const Udt R = slot_ref(h, idx);
                // stack.jl:534-536 wraps greens_temp unconditionally; its result is only observable through the
                // check, so the wrap is skipped when the check is off.  (Slab form: out of place, in front of the
                // slice sequence.)
                const bool wt = h->p.check_propagation_error != 0, wt_slab = wt && h->slab && !h->cb.on;
                CHK(add_slice_sequence_left(h, idx, wt_slab));
                const Udt L = slot_ref(h, idx);
                if (wt && !wt_slab) {
                    CHK(copy_mat(h, h->greens_temp, h->greens));
                    CHK(wrap_greens(h, &h->greens_temp, h->current_slice - 1, 1));
                }
                if (wt_slab) {
                    // the wrap reads mc.s.greens: the new Green's function goes to the other buffer of the pair
                    // (greens_alt, free between two sweep_spatial calls) instead of waiting for it
                    std::swap(h->greens, h->greens_alt);
                    CHK(calculate_greens_src(h, h->greens, L, R));
                    CHK(prop_check(h));
                    return 0;
                }
                CHK(calculate_greens_src(h, h->greens, L, R));
                if (h->p.check_propagation_error) CHK(prop_check(h));
            } else {
                CHK(add_slice_sequence_left(h, h->K));
                h->direction = -1;
                h->current_slice = M + 1;
                CHK(propagate(h));
            }
        } else {
            CHK(wrap_greens(h, &h->greens, h->current_slice, 1));
            h->current_slice += 1;
        }
    } else {
        if ((h->current_slice - 1) % s == 0) {
            h->current_slice -= 1;
            if (h->current_slice == M) {
                const Udt L = slot_ref(h, h->K);
                CHK(reset_slot(h, h->K));
                CHK(calculate_greens_src(h, h->greens, L, slot_ref(h, h->K)));
                CHK(wrap_greens(h, &h->greens, h->current_slice + 1, -1));
            } else if (0 < h->current_slice && h->current_slice < M) {
                const int idx = h->current_slice / s + 1;
                const Udt L = slot_ref(h, idx - 1);
                CHK(add_slice_sequence_right(h, idx));
                const Udt R = slot_ref(h, idx - 1);
                // greens_temp = old Green's function (stack.jl:596-600): exchange the buffers instead of copying
                if (h->p.check_propagation_error) std::swap(h->greens_temp, h->greens);
                CHK(calculate_greens_src(h, h->greens, L, R));
                if (h->p.check_propagation_error) CHK(prop_check(h));
                CHK(wrap_greens(h, &h->greens, h->current_slice + 1, -1));
            } else {
                CHK(add_slice_sequence_right(h, 1));
                h->direction = 1;
                h->current_slice = 0;
                CHK(propagate(h));
            }
        } else {
            CHK(wrap_greens(h, &h->greens, h->current_slice, -1));
            h->current_slice -= 1;
        }
    }
    return 0;
}

// DQMC.jl:546-582
static int sweep_spatial_launches(dqmc_handle *h);
static int sweep_spatial(dqmc_handle *h)
{
    const int l = h->current_slice;
    if (l < 1 || l > h->M) return fail(h, DQMC_ERR_STATE, "sweep_spatial: current_slice outside 1..slices");
    CHK(sweep_spatial_launches(h));
    return 0;
}
static int sweep_spatial_launches(dqmc_handle *h)
{
    const int l = h->current_slice;
    int8_t *cslice = h->conf + (long)(l - 1) * h->N;
    h->conf_version++;
    if (h->sweep_lu) {
        // decide on the 64 x 64 block (four waves per walker), apply the chunk out of place with MFMA
        double *cur = h->greens, *alt = h->greens_alt;
        const size_t istr = (size_t)h->units * sweep_lu_image_doubles();
        const long cstr = (long)h->N * h->M;
        hipEvent_t a, b;
        if (h->sweep_fused && h->n % 64 == 0 && h->N >= 128) {
            // the elimination of chunk c runs beside the flush of chunk c - 1 (one launch per chunk boundary)
            const int nc = h->N / 64;
            timing_events(h, &a, &b);
            HIPCHK(launch_sweep_lu(h->n, h->nb, h->W, cur, h->nn, cslice, cstr, 0, 64, h->lu_img, h->sc, h->rng, h->stats,
                                   h->p.check_sign_problem, h->qr_ws.errflag, h->cur, a, b));
            CHK(timing_push(h, a, b, DQMC_K_SWEEP));
            for (int c = 1; c < nc; ++c) {
                timing_events(h, &a, &b);
                HIPCHK(launch_sweep_fused(h->n, h->nb, h->W, cur, alt, h->nn, cslice, cstr, 64 * c, 64 * (c - 1),
                                          h->lu_img + (size_t)(c & 1) * istr, h->lu_img + (size_t)((c - 1) & 1) * istr, h->sc,
                                          h->rng, h->stats, h->p.check_sign_problem, h->qr_ws.errflag, h->cur, a, b));
                CHK(timing_push(h, a, b, DQMC_K_SWEEP));
                std::swap(cur, alt);
            }
            timing_events(h, &a, &b);
            HIPCHK(launch_sweep_flush_lu(h->n, h->units, cur, alt, h->nn, 64 * (nc - 1), 64,
                                         h->lu_img + (size_t)((nc - 1) & 1) * istr, h->cur, a, b));
            CHK(timing_push(h, a, b, DQMC_K_FLUSH));
            std::swap(cur, alt);
            if (cur != h->greens) std::swap(h->greens, h->greens_alt);
            return 0;
        }
        for (int site0 = 0; site0 < h->N; site0 += 64) {
            const int ns = std::min(64, h->N - site0);
            timing_events(h, &a, &b);
            HIPCHK(launch_sweep_lu(h->n, h->nb, h->W, cur, h->nn, cslice, cstr, site0, ns, h->lu_img, h->sc,
                                   h->rng, h->stats, h->p.check_sign_problem, h->qr_ws.errflag, h->cur, a, b));
            CHK(timing_push(h, a, b, DQMC_K_SWEEP));
            timing_events(h, &a, &b);
            HIPCHK(launch_sweep_flush_lu(h->n, h->units, cur, alt, h->nn, site0, ns, h->lu_img, h->cur, a, b));
            CHK(timing_push(h, a, b, DQMC_K_FLUSH));
            std::swap(cur, alt);
        }
        if (cur != h->greens) std::swap(h->greens, h->greens_alt);
        return 0;
    }
    for (int site0 = 0; site0 < h->N; site0 += h->kd) {
        const int ns = std::min(h->kd, h->N - site0);
        {
            Timed t(h, DQMC_K_SWEEP);
            HIPCHK(launch_sweep_chunk(h->n, h->nb, h->W, h->p.model_kind, h->greens, h->nn, cslice, (long)h->N * h->M,
                                      site0, ns, h->sU, h->sVT, (long)h->n * h->kd, h->sc, h->rng, h->stats,
                                      h->p.check_sign_problem, h->cur));
        }
        if (h->n % 64 == 0 && h->kd == 64) {  // dedicated flush kernel (whole K in LDS, C requested first)
            hipEvent_t a = nullptr, b = nullptr;
            if (h->timing) {
                auto get = [&]() {
                    hipEvent_t ev;
                    if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
                    else (void)hipEventCreate(&ev);
                    return ev;
                };
                a = get(); b = get();
            }
            HIPCHK(launch_gemm_flush(h->n, h->units, h->sU, h->sVT, (long)h->n * h->kd, h->greens, h->nn, h->cur, a, b));
            if (h->timing) {
                h->pending.push_back({a, b, DQMC_K_GEMM});
                if (h->pending.size() >= 2048) CHK(timing_drain(h));
            }
            continue;
        }
        GemmArgs g = gemm_base(h, mat(h->sU, (long)h->n * h->kd, h->n), 0, mat(h->sVT, (long)h->n * h->kd, h->n), 1,
                               h->greens);
        g.K = h->kd;
        g.beta = 1;
        CHK(run_gemm(h, g));
    }
    return 0;
}

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *dqmc_last_error(const dqmc_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dqmc_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int dqmc_create(const dqmc_params *p, dqmc_handle **out)
{
    if (!p || !out) return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: null argument");
    *out = nullptr;
    refresh_kernel_switches();  // (the launchers' A/B switches: read here, never inside a launch)
    if (p->n_sites < 1 || p->slices < 1 || p->safe_mult < 1 || p->n_walkers < 1)
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: n_sites, slices, safe_mult, n_walkers must be >= 1");
    if (p->slices % p->safe_mult != 0)  // stack.jl:115: convert(Int, slices / safe_mult) throws InexactError
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: slices must be divisible by safe_mult");
    if (p->model_kind != DQMC_ATTRACTIVE && p->model_kind != DQMC_REPULSIVE)
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: unknown model_kind");
    if (!p->eT || !p->eTinv || !p->eT2 || !p->eTinv2)
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: hopping exponentials missing");
    if (!(p->U >= 0.0)) return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: U must be positive");
    const int nb = p->model_kind == DQMC_REPULSIVE ? 2 : 1;
    const bool sweep_old = getenv("DQMC_SWEEP_OLD") != nullptr;  // one-workgroup-per-walker chunk kernel (A/B only)
    if (p->n_sites > 1024) return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: n_sites exceeds 1024 (unsupported)");
    if (sweep_old && nb * ((p->n_sites + 63) & ~63) > 1024)
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: DQMC_SWEEP_OLD needs n_blocks * n_sites <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, DQMC_ERR_NO_DEVICE, "dqmc_create: no HIP device visible");
    if (p->device_id < 0 || p->device_id >= ndev)
        return fail(nullptr, DQMC_ERR_INVALID, "dqmc_create: device_id out of range");

    dqmc_handle *h = new dqmc_handle();
    h->p = *p;
    h->p.eT = h->p.eTinv = h->p.eT2 = h->p.eTinv2 = nullptr;
    h->N = h->n = p->n_sites;
    h->nb = nb;
    h->M = p->slices;
    h->s = p->safe_mult;
    h->K = h->M / h->s;
    h->W = p->n_walkers;
    h->units = h->W * h->nb;
    h->nn = (long)h->n * h->n;
    h->kd = sweep_kd(h->n, h->nb);
    h->sweep_lu = !sweep_old;
    // elimination of chunk c beside the flush of chunk c - 1 in one launch (the elimination first applies the previous
    // chunk to its own 64 x 64 block; the flush workgroups take two column passes each, so that at 32 units the whole
    // launch is co-resident, one workgroup per CU): 36 us per chunk against 24 + 17 for the two separate launches.
    // DQMC_SWEEP_SPLIT selects the separate launches.
    // With more units than that the phase is throughput-bound and the separate launches win (config 4 on one GPU,
    // 512 units: 688 vs 715 ms per sweep), so the fused form is used only when its grid fits the CUs.
    h->sweep_fused = false;
    if (getenv("DQMC_SWEEP_SPLIT") == nullptr && h->n % 64 == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, p->device_id) == hipSuccess) {
            const int ncp = (h->n % 256 == 0) ? 2 : 1, nt = (h->n % 128 == 0) ? 8 : 4;
            const int flush_blocks = ((h->units + 7) / 8) * 8 * (h->n / 64) * (h->n / (16 * nt * ncp));
            h->sweep_fused = h->W + flush_blocks <= prop.multiProcessorCount;
        }
    }
    // lambda = acosh(exp(U*dtau/2)) (Attractive.jl:103,118; Repulsive.jl:116,138)
    h->lambda = std::acosh(std::exp(0.5 * p->U * p->delta_tau));
    h->epl = std::exp(h->lambda);
    h->eml = std::exp(-h->lambda);
    for (int ci = 0; ci < 2; ++ci) {
        const double c = ci ? 1.0 : -1.0;
        const double dE = -2.0 * h->lambda * c;
        h->sc.gamma[ci] = std::exp(dE) - 1.0;
        h->sc.ebos[ci] = std::exp(-dE);
        h->sc.dup[ci] = std::exp(dE) - 1.0;
        h->sc.ddn[ci] = std::exp(-dE) - 1.0;
    }
    auto bail = [&](int rc) {
        g_create_error = h->err;
        dqmc_destroy(h);
        return rc;
    };
#define CCHK(expr)                     \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != 0) return bail(rc__); \
    } while (0)
#define CHIP(expr)                                                        \
    do {                                                                  \
        hipError_t e__ = (expr);                                          \
        if (e__ != hipSuccess) {                                          \
            h->err = std::string(#expr) + ": " + hipGetErrorString(e__);  \
            return bail(DQMC_ERR_HIP);                                    \
        }                                                                 \
    } while (0)
    CHIP(hipSetDevice(p->device_id));
    CHIP(hipStreamCreate(&h->stream));
    h->cur = h->stream;
    const size_t cn = (size_t)nb * h->nn, un = (size_t)h->units * h->nn, uv = (size_t)h->units * h->n;
    CCHK(dalloc(h, &h->eT, cn)); CCHK(dalloc(h, &h->eTinv, cn));
    CCHK(dalloc(h, &h->eT2, cn)); CCHK(dalloc(h, &h->eTinv2, cn));
    CHIP(hipMemcpy(h->eT, p->eT, cn * sizeof(double), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(h->eTinv, p->eTinv, cn * sizeof(double), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(h->eT2, p->eT2, cn * sizeof(double), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(h->eTinv2, p->eTinv2, cn * sizeof(double), hipMemcpyHostToDevice));
    if (h->n == 256 && !getenv("DQMC_NO_SLAB")) {
        std::vector<double> tr(cn);
        for (int b = 0; b < nb; ++b)
            for (int j = 0; j < h->n; ++j)
                for (int i = 0; i < h->n; ++i) tr[(size_t)b * h->nn + j + (size_t)h->n * i] = p->eT2[(size_t)b * h->nn + i + (size_t)h->n * j];
        CCHK(dalloc(h, &h->eT2T, cn));
        CHIP(hipMemcpy(h->eT2T, tr.data(), cn * sizeof(double), hipMemcpyHostToDevice));
        h->slab = true;
    }
    CCHK(dalloc(h, &h->conf, (size_t)h->W * h->N * h->M));
    {
        std::vector<int8_t> ones((size_t)h->W * h->N * h->M, 1);
        CHIP(hipMemcpy(h->conf, ones.data(), ones.size(), hipMemcpyHostToDevice));
    }
    CCHK(dalloc(h, &h->u_stack, (size_t)(h->K + 2) * un));
    CCHK(dalloc(h, &h->t_stack, (size_t)(h->K + 2) * un));
    CCHK(dalloc(h, &h->d_stack, (size_t)(h->K + 2) * uv));
    for (int i = 0; i <= h->K + 1; ++i) {
        h->su.push_back(h->u_stack + (size_t)i * un);
        h->st.push_back(h->t_stack + (size_t)i * un);
        h->sd.push_back(h->d_stack + (size_t)i * uv);
    }
    double **mats[] = {&h->Ul, &h->Ur, &h->Tl, &h->Tr, &h->greens, &h->greens_temp, &h->tmp1,
                       &h->tmp2, &h->bufA, &h->bufB, &h->qrV, &h->qrW, &h->qrS};
    for (auto m : mats) CCHK(dalloc(h, m, un));
    CCHK(dalloc(h, &h->Dl, uv)); CCHK(dalloc(h, &h->Dr, uv)); CCHK(dalloc(h, &h->tau, uv));
    const size_t wn = (size_t)h->units * ((h->n + 15) / 16) * 256;
    CCHK(dalloc(h, &h->trsm_w, wn));
    if (h->n > 256) CCHK(dalloc(h, &h->trsm_s, un));
    CCHK(dalloc(h, &h->pivot, uv));
    CCHK(alloc_qr_workspace(h));
    CCHK(dalloc(h, &h->sU, (size_t)h->units * h->n * h->kd));
    CCHK(dalloc(h, &h->sVT, (size_t)h->units * h->n * h->kd));
    CCHK(dalloc(h, &h->greens_alt, un));
    CCHK(dalloc(h, &h->lu_img, 2 * (size_t)h->units * sweep_lu_image_doubles()));
    CCHK(dalloc(h, &h->rng, (size_t)h->W));
    CCHK(dalloc(h, &h->stats, (size_t)h->W));
    CCHK(dalloc(h, &h->pc_scratch, 2 * (size_t)h->W));
    {
        std::vector<DevStats> st(h->W);
        for (auto &x : st) {
            x.prop_local = x.acc_local = 0;
            x.negative_probability = {-INFINITY, INFINITY, 0.0, 0};
            x.propagation_error = {-INFINITY, INFINITY, 0.0, 0};
        }
        CHIP(hipMemcpy(h->stats, st.data(), sizeof(DevStats) * h->W, hipMemcpyHostToDevice));
        std::vector<WalkerRng> rg(h->W);
        for (int w = 0; w < h->W; ++w) rg[w] = {(unsigned long long)w, 0ull, nullptr, 0ull, 0};
        CHIP(hipMemcpy(h->rng, rg.data(), sizeof(WalkerRng) * h->W, hipMemcpyHostToDevice));
    }
    h->uniforms.assign(h->W, nullptr);
    h->acc_n = 2 * cn + (size_t)nb * h->n + 1;
    CCHK(dalloc(h, &h->acc, h->acc_n));
    CCHK(init_stack(h));
    CHIP(hipStreamSynchronize(h->stream));
#undef CCHK
#undef CHIP
    *out = h;
    return DQMC_OK;
}

static void ut_free(dqmc_handle *h);
static int ut_reset_accumulators(dqmc_handle *h);
int dqmc_destroy(dqmc_handle *h)
{
    if (!h) return DQMC_OK;
    (void)hipSetDevice(h->p.device_id);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &e : h->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (void *q : h->allocs) (void)hipFree(q);
    for (double *u : h->uniforms)
        if (u) (void)hipFree(u);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    ut_free(h);
    delete h;
    return DQMC_OK;
}

#define ENTER(h)                                                        \
    if (!(h)) return DQMC_ERR_INVALID;                                  \
    HIPCHK(hipSetDevice((h)->p.device_id))
#define WALKER_OK(h, w) \
    if ((w) < 0 || (w) >= (h)->W) return fail((h), DQMC_ERR_INVALID, "walker index out of range")

int dqmc_set_conf(dqmc_handle *h, int32_t w, const int8_t *conf)
{
    ENTER(h); WALKER_OK(h, w);
    const size_t sz = (size_t)h->N * h->M;
    for (size_t i = 0; i < sz; ++i)
        if (conf[i] != 1 && conf[i] != -1) return fail(h, DQMC_ERR_INVALID, "conf entries must be +1 or -1");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->conf + (size_t)w * sz, conf, sz, hipMemcpyHostToDevice));
    h->conf_version++;
    return DQMC_OK;
}
int dqmc_get_conf(dqmc_handle *h, int32_t w, int8_t *conf)
{
    ENTER(h); WALKER_OK(h, w);
    const size_t sz = (size_t)h->N * h->M;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(conf, h->conf + (size_t)w * sz, sz, hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_set_uniforms(dqmc_handle *h, int32_t w, const double *u, size_t n)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->uniforms[w]) { HIPCHK(hipFree(h->uniforms[w])); h->uniforms[w] = nullptr; }
    HIPCHK(hipMalloc((void **)&h->uniforms[w], (n ? n : 1) * sizeof(double)));
    HIPCHK(hipMemcpy(h->uniforms[w], u, n * sizeof(double), hipMemcpyHostToDevice));
    WalkerRng r = {0ull, 0ull, h->uniforms[w], (unsigned long long)n, 0};
    HIPCHK(hipMemcpy(h->rng + w, &r, sizeof(r), hipMemcpyHostToDevice));
    return DQMC_OK;
}
int dqmc_seed(dqmc_handle *h, int32_t w, uint64_t seed)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    WalkerRng r = {(unsigned long long)seed, 0ull, nullptr, 0ull, 0};
    HIPCHK(hipMemcpy(h->rng + w, &r, sizeof(r), hipMemcpyHostToDevice));
    return DQMC_OK;
}
int dqmc_uniforms_used(dqmc_handle *h, int32_t w, uint64_t *used)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    WalkerRng r;
    HIPCHK(hipMemcpy(&r, h->rng + w, sizeof(r), hipMemcpyDeviceToHost));
    *used = r.draw;
    return DQMC_OK;
}
int dqmc_get_state(dqmc_handle *h, int32_t *cs, int32_t *dir)
{
    if (!h) return DQMC_ERR_INVALID;
    if (cs) *cs = h->current_slice;
    if (dir) *dir = h->direction;
    return DQMC_OK;
}

static int check_rng(dqmc_handle *h)
{
    std::vector<WalkerRng> rg(h->W);
    HIPCHK(hipMemcpy(rg.data(), h->rng, sizeof(WalkerRng) * h->W, hipMemcpyDeviceToHost));
    for (int w = 0; w < h->W; ++w)
        if (rg[w].exhausted) return fail(h, DQMC_ERR_RNG, "host-supplied uniform stream exhausted");
    return 0;
}
int dqmc_synchronize(dqmc_handle *h)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(timing_drain(h));
    CHK(check_qr_workspace(h));
    return check_rng(h);
}
int dqmc_build_stack(dqmc_handle *h)
{
    ENTER(h);
    CHK(build_stack(h));
    h->prepared = true;
    return dqmc_synchronize(h);
}
int dqmc_prepare(dqmc_handle *h)
{
    ENTER(h);
    CHK(init_stack(h));
    CHK(build_stack(h));
    CHK(propagate(h));
    h->prepared = true;
    return dqmc_synchronize(h);
}
#define NEED_PREPARED(h) \
    if (!(h)->prepared) return fail((h), DQMC_ERR_STATE, "call dqmc_prepare or dqmc_build_stack first")
int dqmc_propagate(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    CHK(propagate(h));
    return dqmc_synchronize(h);
}
int dqmc_sweep_spatial(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    CHK(sweep_spatial(h));
    return dqmc_synchronize(h);
}
int dqmc_update(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    CHK(propagate(h));
    CHK(sweep_spatial(h));
    return dqmc_synchronize(h);
}
int dqmc_sweep(dqmc_handle *h, int32_t n_sweeps)
{
    ENTER(h); NEED_PREPARED(h);
    for (int i = 0; i < n_sweeps; ++i)
        for (int u = 0; u < 2 * h->M; ++u) {
            CHK(propagate(h));
            CHK(sweep_spatial(h));
        }
    return dqmc_synchronize(h);
}
int dqmc_update_until_measure(dqmc_handle *h, int32_t *n_updates)
{
    ENTER(h); NEED_PREPARED(h);
    int cnt = 0;
    do {
        CHK(propagate(h));
        CHK(sweep_spatial(h));
        ++cnt;
    } while (!(h->current_slice == 1 && h->direction == 1));
    if (n_updates) *n_updates = cnt;
    return dqmc_synchronize(h);
}

int dqmc_get_greens_eff(dqmc_handle *h, int32_t w, double *out)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->greens + (size_t)w * h->nb * h->nn, sizeof(double) * h->nb * h->nn, hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_set_greens_eff(dqmc_handle *h, int32_t w, const double *in)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->greens + (size_t)w * h->nb * h->nn, in, sizeof(double) * h->nb * h->nn, hipMemcpyHostToDevice));
    return DQMC_OK;
}
// greens!(mc): temp = greens*eT; out = eTinv*temp (DQMC.jl:721-730); result in tmp2
static int true_greens(dqmc_handle *h, const double *src)
{
    if (h->cb.on) {
        CHK(cb_mult(h, CB_RIGHT_ET, 0, src, h->tmp1, nullptr));
        CHK(cb_mult(h, CB_LEFT_ETINV, 0, h->tmp1, h->tmp2, nullptr));
        return 0;
    }
    CHK(run_gemm(h, gemm_base(h, U_(h, src), 0, C_(h, h->eT), 0, h->tmp1)));
    CHK(run_gemm(h, gemm_base(h, C_(h, h->eTinv), 0, U_(h, h->tmp1), 0, h->tmp2)));
    return 0;
}
int dqmc_get_greens(dqmc_handle *h, int32_t w, double *out)
{
    ENTER(h); WALKER_OK(h, w);
    CHK(true_greens(h, h->greens));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->tmp2 + (size_t)w * h->nb * h->nn, sizeof(double) * h->nb * h->nn, hipMemcpyDeviceToHost));
    return DQMC_OK;
}

// calculate_greens(mc, slice, output) (stack.jl:422-480) for every walker of the handle
static int calculate_greens_from_scratch(dqmc_handle *h, int slice, double *output)
{
    const int n = h->n, M = h->M, s = h->s;
    // right factor: Ur,Dr,Tr = B(slice+1)' ... B(M)'
    CHK(set_identity(h, h->bufA)); CHK(set_identity(h, h->Ur)); CHK(set_ones(h, h->Dr)); CHK(set_identity(h, h->Tr));
    double *cur = h->bufA, *oth = h->bufB;
    auto chain_step = [&](int k, bool dagger, bool stab, double *D, double *T, double *Ufinal) -> int {
        if (h->cb.on) CHK(cb_mult(h, dagger ? CB_LEFT_BDAG : CB_LEFT_B, k, cur, oth, stab ? D : nullptr));
        else {
            GemmArgs g = dagger ? gemm_base(h, C_(h, h->eT2), 1, U_(h, cur), 0, oth)
                                : gemm_base(h, C_(h, h->eT2), 0, U_(h, cur), 0, oth);
            if (dagger) { g.rowscale = vs_conf(h, k, +1); g.row_first = 1; }
            else g.kscale = vs_conf(h, k, +1);
            if (stab) g.colscale = vs_arr(D, n);
            CHK(run_gemm(h, g));
        }
        std::swap(cur, oth);
        if (stab) {
            // udt(curr_U, D, tmp1); T = tmp1 * T  (stack.jl:441-444)
            CHK(udt(h, cur, Ufinal ? Ufinal : oth, D, h->tmp2, 1));
            if (!Ufinal) std::swap(cur, oth);
            CHK(copy_mat(h, h->tmp1, T));
            CHK(run_gemm(h, gemm_base(h, U_(h, h->tmp2), 0, U_(h, h->tmp1), 0, T)));
        }
        return 0;
    };
    if (slice + 1 <= M) {
        for (int k = M; k >= slice + 1; --k) CHK(chain_step(k, true, k % s == 0, h->Dr, h->Tr, nullptr));
        // final: tmp1 = curr_U*Diagonal(Dr); udt(Ur, Dr, tmp1); Tr = tmp1*Tr
        GemmArgs g = gemm_base(h, U_(h, cur), 0, U_(h, h->qrS), 0, oth);
        CHK(set_identity(h, h->qrS));
        g.colscale = vs_arr(h->Dr, n);
        CHK(run_gemm(h, g));
        std::swap(cur, oth);
        CHK(udt(h, cur, h->Ur, h->Dr, h->tmp2, 1));
        CHK(copy_mat(h, h->tmp1, h->Tr));
        CHK(run_gemm(h, gemm_base(h, U_(h, h->tmp2), 0, U_(h, h->tmp1), 0, h->Tr)));
    }
    // left factor: Ul,Dl,Tl = B(slice) ... B(1)
    CHK(set_identity(h, h->bufA)); CHK(set_identity(h, h->Ul)); CHK(set_ones(h, h->Dl)); CHK(set_identity(h, h->Tl));
    cur = h->bufA; oth = h->bufB;
    if (slice >= 1) {
        for (int k = 1; k <= slice; ++k) CHK(chain_step(k, false, k % s == 0, h->Dl, h->Tl, nullptr));
        GemmArgs g = gemm_base(h, U_(h, cur), 0, U_(h, h->qrS), 0, oth);
        CHK(set_identity(h, h->qrS));
        g.colscale = vs_arr(h->Dl, n);
        CHK(run_gemm(h, g));
        std::swap(cur, oth);
        CHK(udt(h, cur, h->Ul, h->Dl, h->tmp2, 1));
        CHK(copy_mat(h, h->tmp1, h->Tl));
        CHK(run_gemm(h, gemm_base(h, U_(h, h->tmp2), 0, U_(h, h->tmp1), 0, h->Tl)));
    }
    CHK(calculate_greens(h, output));
    return 0;
}
int dqmc_calculate_greens_at(dqmc_handle *h, int32_t w, int32_t slice, double *out)
{
    ENTER(h); WALKER_OK(h, w);
    if (slice < 0 || slice > h->M) return fail(h, DQMC_ERR_INVALID, "slice out of range 0..slices");
    CHK(calculate_greens_from_scratch(h, slice, h->greens_temp));
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(check_qr_workspace(h));
    HIPCHK(hipMemcpy(out, h->greens_temp + (size_t)w * h->nb * h->nn, sizeof(double) * h->nb * h->nn,
                     hipMemcpyDeviceToHost));
    return DQMC_OK;
}
// the per-configuration step of replay!(mc) (DQMC.jl:651-653): calculate_greens(mc, slice) into
// mc.s.greens for every walker; current_slice is set like replay! does (DQMC.jl:647)
int dqmc_replay_greens(dqmc_handle *h, int32_t slice)
{
    ENTER(h);
    if (slice < 0 || slice > h->M) return fail(h, DQMC_ERR_INVALID, "slice out of range 0..slices");
    CHK(calculate_greens_from_scratch(h, slice, h->greens));
    h->current_slice = 1;
    h->prepared = true;
    return dqmc_synchronize(h);
}
// compress(mc, model, conf) / decompress (HubbardModel.jl:56-59): Julia BitArray chunks
int dqmc_get_conf_bits(dqmc_handle *h, int32_t w, uint64_t *chunks)
{
    ENTER(h); WALKER_OK(h, w);
    const size_t sz = (size_t)h->N * h->M, nch = (sz + 63) / 64;
    unsigned long long *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, nch * sizeof(unsigned long long)));
    hipError_t e1 = launch_conf_pack(h->conf + (size_t)w * sz, sz, d, h->stream);
    hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(h->stream) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipMemcpy(chunks, d, nch * sizeof(unsigned long long), hipMemcpyDeviceToHost) : e2;
    (void)hipFree(d);
    HIPCHK(e3);
    return DQMC_OK;
}
int dqmc_set_conf_bits(dqmc_handle *h, int32_t w, const uint64_t *chunks)
{
    ENTER(h); WALKER_OK(h, w);
    const size_t sz = (size_t)h->N * h->M, nch = (sz + 63) / 64;
    unsigned long long *d = nullptr;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMalloc((void **)&d, nch * sizeof(unsigned long long)));
    hipError_t e1 = hipMemcpy(d, chunks, nch * sizeof(unsigned long long), hipMemcpyHostToDevice);
    hipError_t e2 = e1 == hipSuccess ? launch_conf_unpack(d, sz, h->conf + (size_t)w * sz, h->stream) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipStreamSynchronize(h->stream) : e2;
    (void)hipFree(d);
    HIPCHK(e3);
    h->conf_version++;
    return DQMC_OK;
}

int dqmc_wrap_greens(dqmc_handle *h, int32_t slice, int32_t direction)
{
    ENTER(h);
    if (direction != 1 && direction != -1) return fail(h, DQMC_ERR_INVALID, "direction must be +1 or -1");
    const int l = direction == 1 ? slice : slice - 1;
    if (l < 1 || l > h->M) return fail(h, DQMC_ERR_INVALID, "wrap_greens: slice out of range");
    CHK(wrap_greens(h, &h->greens, slice, direction));
    return dqmc_synchronize(h);
}

static void conv_mag(const DevMagStats &d, dqmc_magstats &o) { o.max = d.max; o.min = d.min; o.sum = d.sum; o.count = d.count; }
int dqmc_get_stats(dqmc_handle *h, int32_t w, dqmc_stats *out)
{
    ENTER(h); WALKER_OK(h, w);
    HIPCHK(hipStreamSynchronize(h->stream));
    DevStats d;
    HIPCHK(hipMemcpy(&d, h->stats + w, sizeof(d), hipMemcpyDeviceToHost));
    out->prop_local = d.prop_local;
    out->acc_local = d.acc_local;
    out->imaginary_probability = {-INFINITY, INFINITY, 0.0, 0};
    conv_mag(d.negative_probability, out->negative_probability);
    conv_mag(d.propagation_error, out->propagation_error);
    return DQMC_OK;
}

int dqmc_accumulate_greens(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    CHK(true_greens(h, h->greens));
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_accumulate(h->n, h->nb, h->W, h->tmp2, h->nn, h->acc, h->stream));
    }
    return DQMC_OK;
}
// EachSitePairByDistance(lattice) (lattice_iterators.jl:157-190) as a direction table
int dqmc_set_pair_directions(dqmc_handle *h, const int32_t *dir_of, int32_t n_dirs)
{
    ENTER(h);
    const int n = h->n;
    if (!dir_of || n_dirs < 1 || n_dirs > n * n) return fail(h, DQMC_ERR_INVALID, "bad direction table");
    std::vector<int> ptr(n_dirs + 1, 0), src((size_t)n * n), trg((size_t)n * n);
    for (int s = 0; s < n; ++s)
        for (int t = 0; t < n; ++t) {
            const int d = dir_of[s + (size_t)n * t];
            if (d < 0 || d >= n_dirs) return fail(h, DQMC_ERR_INVALID, "direction index out of range");
            ptr[d + 1]++;
        }
    for (int d = 0; d < n_dirs; ++d) ptr[d + 1] += ptr[d];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int s = 0; s < n; ++s)          // the reference's order inside a direction: src outer, trg inner
        for (int t = 0; t < n; ++t) {
            const int d = dir_of[s + (size_t)n * t];
            src[fill[d]] = s;
            trg[fill[d]] = t;
            fill[d]++;
        }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!h->pair_src) {
        CHK(dalloc(h, &h->pair_src, (size_t)n * n));
        CHK(dalloc(h, &h->pair_trg, (size_t)n * n));
    }
    CHK(dalloc(h, &h->dir_ptr, (size_t)n_dirs + 1));
    h->n_dirs = n_dirs;
    h->corr_n = 4 * (size_t)n_dirs + 3 * (size_t)n + 1;
    h->red_valid = false;  // (re)sized: the last reduction is void
    CHK(dalloc(h, &h->corr_per_walker, (size_t)h->W * 4 * n_dirs));
    CHK(dalloc(h, &h->corr_acc, h->corr_n));
    HIPCHK(hipMemcpy(h->dir_ptr, ptr.data(), sizeof(int) * (n_dirs + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->pair_src, src.data(), sizeof(int) * n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->pair_trg, trg.data(), sizeof(int) * n * n, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
int dqmc_accumulate_correlations(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    CHK(true_greens(h, h->greens));
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_correlations(h->n, h->nb, h->p.model_kind, h->W, h->tmp2, h->nn, h->dir_ptr, h->pair_src,
                                   h->pair_trg, h->n_dirs, h->corr_per_walker, h->corr_acc, h->stream));
    }
    return DQMC_OK;
}
int dqmc_correlations_size(dqmc_handle *h, size_t *n)
{
    if (!h || !n) return DQMC_ERR_INVALID;
    *n = h->corr_n;
    return DQMC_OK;
}
int dqmc_get_correlations(dqmc_handle *h, double *host_out)
{
    ENTER(h);
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, h->corr_acc, h->corr_n * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_export_correlations(dqmc_handle *h, void *device_out)
{
    ENTER(h);
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    HIPCHK(hipMemcpyAsync(device_out, h->corr_acc, h->corr_n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
// EachLocalQuadByDistance{K}(lattice) (lattice_iterators.jl:264-318) as a target table
int dqmc_set_local_targets(dqmc_handle *h, const int32_t *trg_of, int32_t K)
{
    ENTER(h);
    const int n = h->n;
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    if (!trg_of || K < 1 || K > h->n_dirs) return fail(h, DQMC_ERR_INVALID, "bad target table");
    for (size_t i = 0; i < (size_t)n * K; ++i)
        if (trg_of[i] < -1 || trg_of[i] >= n) return fail(h, DQMC_ERR_INVALID, "target index out of range");
    HIPCHK(hipStreamSynchronize(h->stream));
    h->K_loc = K;
    h->pc_n = (size_t)h->n_dirs * K * K + 1;
    h->red_valid = false;
    CHK(dalloc(h, &h->trg_of, (size_t)n * K));
    CHK(dalloc(h, &h->pc_per_walker, (size_t)h->W * (h->pc_n - 1)));
    CHK(dalloc(h, &h->pc_acc, h->pc_n));
    HIPCHK(hipMemcpy(h->trg_of, trg_of, sizeof(int) * n * K, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
int dqmc_accumulate_pairing(dqmc_handle *h)
{
    ENTER(h); NEED_PREPARED(h);
    if (!h->K_loc) return fail(h, DQMC_ERR_STATE, "call dqmc_set_local_targets first");
    CHK(true_greens(h, h->greens));
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_pairing(h->n, h->nb, h->W, h->tmp2, h->nn, h->dir_ptr, h->pair_src, h->pair_trg, h->n_dirs,
                              h->K_loc, h->trg_of, h->pc_per_walker, h->pc_acc, h->stream));
    }
    return DQMC_OK;
}
int dqmc_pairing_size(dqmc_handle *h, size_t *n)
{
    if (!h || !n) return DQMC_ERR_INVALID;
    *n = h->pc_n;
    return DQMC_OK;
}
int dqmc_get_pairing(dqmc_handle *h, double *host_out)
{
    ENTER(h);
    if (!h->K_loc) return fail(h, DQMC_ERR_STATE, "call dqmc_set_local_targets first");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, h->pc_acc, h->pc_n * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_export_pairing(dqmc_handle *h, void *device_out)
{
    ENTER(h);
    if (!h->K_loc) return fail(h, DQMC_ERR_STATE, "call dqmc_set_local_targets first");
    HIPCHK(hipMemcpyAsync(device_out, h->pc_acc, h->pc_n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
int dqmc_accumulator_size(dqmc_handle *h, size_t *n)
{
    if (!h || !n) return DQMC_ERR_INVALID;
    *n = h->acc_n;
    return DQMC_OK;
}
int dqmc_reset_accumulators(dqmc_handle *h)
{
    ENTER(h);
    h->red_valid = false;  // (the last reduction no longer describes the accumulators)
    HIPCHK(hipMemsetAsync(h->acc, 0, h->acc_n * sizeof(double), h->stream));
    if (h->corr_acc) HIPCHK(hipMemsetAsync(h->corr_acc, 0, h->corr_n * sizeof(double), h->stream));
    if (h->pc_acc) HIPCHK(hipMemsetAsync(h->pc_acc, 0, h->pc_n * sizeof(double), h->stream));
    CHK(ut_reset_accumulators(h));
    return DQMC_OK;
}
int dqmc_get_accumulators(dqmc_handle *h, double *host_out)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, h->acc, h->acc_n * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_export_accumulators(dqmc_handle *h, void *device_out)
{
    ENTER(h);
    HIPCHK(hipMemcpyAsync(device_out, h->acc, h->acc_n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}

#include "unequal_time.inl"

// ---------------------------------------------------------------------------
// Measurement reduction over ranks: every accumulator the handle keeps and the DQMCAnalysis
// counters, as ONE packed device buffer of doubles [sums | maxima | minima]:
//   sums   = acc (G, G.^2, occupation, count), correlations, pairing, susceptibilities (each if configured),
//            then prop_local, acc_local, negative_probability {sum, count}, propagation_error {sum, count}
//   maxima = negative_probability.max, propagation_error.max      minima = the two .min
// (MagnitudeStats, DQMC.jl:4-47).  dqmc_reduce runs three ncclAllReduce (sum / max / min) on the handle's stream;
// a host-side collective (MPI from Julia, gloo in the tests) can do the same through export / import.
static const size_t RED_STAT_SUMS = 6;
static size_t red_nsum(dqmc_handle *h)
{
    return h->acc_n + h->corr_n + h->pc_n + (h->ut ? h->ut->sus_n : 0) + RED_STAT_SUMS;
}
static int red_pack(dqmc_handle *h)
{
    const size_t nsum = red_nsum(h), tot = nsum + 4;
    if (h->red_cap < tot) {
        CHK(dalloc(h, &h->red_buf, tot));
        h->red_cap = tot;
    }
    size_t off = 0;
    auto put = [&](const double *src, size_t cnt) -> int {
        if (cnt) HIPCHK(hipMemcpyAsync(h->red_buf + off, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        off += cnt;
        return 0;
    };
    CHK(put(h->acc, h->acc_n));
    CHK(put(h->corr_acc, h->corr_n));
    CHK(put(h->pc_acc, h->pc_n));
    if (h->ut) CHK(put(h->ut->sus_acc, h->ut->sus_n));
    h->red_sizes[0] = h->acc_n; h->red_sizes[1] = h->corr_n; h->red_sizes[2] = h->pc_n; h->red_sizes[3] = h->ut ? h->ut->sus_n : 0;
    // counters of the local walkers, reduced on the host in walker order
    std::vector<DevStats> st(h->W);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(st.data(), h->stats, sizeof(DevStats) * h->W, hipMemcpyDeviceToHost));
    double tail[RED_STAT_SUMS + 4] = {0, 0, 0, 0, 0, 0, -INFINITY, -INFINITY, INFINITY, INFINITY};
    for (const auto &x : st) {
        tail[0] += (double)x.prop_local;
        tail[1] += (double)x.acc_local;
        tail[2] += x.negative_probability.sum;
        tail[3] += (double)x.negative_probability.count;
        tail[4] += x.propagation_error.sum;
        tail[5] += (double)x.propagation_error.count;
        tail[6] = std::fmax(tail[6], x.negative_probability.max);
        tail[7] = std::fmax(tail[7], x.propagation_error.max);
        tail[8] = std::fmin(tail[8], x.negative_probability.min);
        tail[9] = std::fmin(tail[9], x.propagation_error.min);
    }
    HIPCHK(hipMemcpy(h->red_buf + off, tail, sizeof(tail), hipMemcpyHostToDevice));
    return 0;
}
// The reduced sums stay in red_buf (read with dqmc_get_reduced): the handle's own accumulators keep the LOCAL sums, so
// that the reduction can be repeated every measure_rate sweeps (DQMC.jl:429-436) - writing the global sums back into
// them would count the earlier samples once per rank again at the next reduction.
static int red_unpack(dqmc_handle *h)
{
    const size_t off = red_nsum(h) - RED_STAT_SUMS;
    double tail[RED_STAT_SUMS + 4];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(tail, h->red_buf + off, sizeof(tail), hipMemcpyDeviceToHost));
    dqmc_stats &r = h->red_stats;
    r.prop_local = (int64_t)std::llround(tail[0]);
    r.acc_local = (int64_t)std::llround(tail[1]);
    r.imaginary_probability = {-INFINITY, INFINITY, 0.0, 0};
    r.negative_probability = {tail[6], tail[8], tail[2], (int64_t)std::llround(tail[3])};
    r.propagation_error = {tail[7], tail[9], tail[4], (int64_t)std::llround(tail[5])};
    h->red_valid = true;
    return 0;
}
int dqmc_comm_unique_id(void *id128)
{
    if (!id128) return DQMC_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail(nullptr, DQMC_ERR_HIP, "ncclGetUniqueId failed");
    std::memcpy(id128, &id, sizeof(id));
    return DQMC_OK;
}
int dqmc_comm_init(const void *id128, int32_t nranks, int32_t rank, int32_t device_id, dqmc_comm **out)
{
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, DQMC_ERR_INVALID, "dqmc_comm_init: bad arguments");
    *out = nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, DQMC_ERR_NO_DEVICE, "dqmc_comm_init: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    dqmc_comm *c = new dqmc_comm();
    c->nranks = nranks; c->rank = rank; c->device = device_id;
    const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(nullptr, DQMC_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    *out = c;
    return DQMC_OK;
}
int dqmc_comm_destroy(dqmc_comm *c)
{
    if (!c) return DQMC_OK;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return DQMC_OK;
}
int dqmc_reduce_size(dqmc_handle *h, size_t *n_doubles)
{
    if (!h || !n_doubles) return DQMC_ERR_INVALID;
    *n_doubles = red_nsum(h) + 4;
    return DQMC_OK;
}
int dqmc_reduce(dqmc_handle *h, dqmc_comm *comm)
{
    ENTER(h);
    CHK(red_pack(h));
    if (comm && comm->comm) {
        if (comm->device != h->p.device_id) return fail(h, DQMC_ERR_INVALID, "dqmc_reduce: communicator bound to another device");
        const size_t nsum = red_nsum(h);
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclAllReduce(h->red_buf, h->red_buf, nsum, ncclDouble, ncclSum, comm->comm, h->stream);
        if (r == ncclSuccess) r = ncclAllReduce(h->red_buf + nsum, h->red_buf + nsum, 2, ncclDouble, ncclMax, comm->comm, h->stream);
        if (r == ncclSuccess) r = ncclAllReduce(h->red_buf + nsum + 2, h->red_buf + nsum + 2, 2, ncclDouble, ncclMin, comm->comm, h->stream);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return fail(h, DQMC_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
    }
    CHK(red_unpack(h));
    return DQMC_OK;
}
// host-mediated variant: pack -> host buffer (reduce it with any collective: sums first, then 2 maxima, 2 minima)
int dqmc_reduce_export(dqmc_handle *h, double *host_out)
{
    ENTER(h);
    CHK(red_pack(h));
    HIPCHK(hipMemcpy(host_out, h->red_buf, (red_nsum(h) + 4) * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_reduce_import(dqmc_handle *h, const double *host_in)
{
    ENTER(h);
    if (h->red_cap < red_nsum(h) + 4) return fail(h, DQMC_ERR_STATE, "call dqmc_reduce_export first");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->red_buf, host_in, (red_nsum(h) + 4) * sizeof(double), hipMemcpyHostToDevice));
    CHK(red_unpack(h));
    return DQMC_OK;
}
// section `which` of the last reduction: 0 Green's-function sums (dqmc_accumulator_size doubles), 1 correlations,
// 2 pairing, 3 susceptibilities (sizes as the local getters report)
int dqmc_get_reduced(dqmc_handle *h, int32_t which, double *host_out)
{
    ENTER(h);
    if (!host_out || which < 0 || which > 3) return fail(h, DQMC_ERR_INVALID, "dqmc_get_reduced: bad arguments");
    if (!h->red_valid) return fail(h, DQMC_ERR_STATE, "call dqmc_reduce first");
    const size_t sizes[4] = {h->acc_n, h->corr_n, h->pc_n, h->ut ? h->ut->sus_n : 0};
    size_t off = 0;
    for (int i = 0; i < which; ++i) off += sizes[i];
    if (sizes[which] == 0) return fail(h, DQMC_ERR_STATE, "dqmc_get_reduced: this accumulator is not configured");
    for (int i = 0; i < 4; ++i)  // (sizes as they are NOW against the sizes that were packed)
        if (sizes[i] != h->red_sizes[i])
            return fail(h, DQMC_ERR_STATE, "accumulators were reconfigured after the last reduction: call dqmc_reduce again");
    if (h->red_cap < red_nsum(h) + 4) return fail(h, DQMC_ERR_STATE, "call dqmc_reduce first");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, h->red_buf + off, sizes[which] * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_get_reduced_stats(dqmc_handle *h, dqmc_stats *out)
{
    if (!h || !out) return DQMC_ERR_INVALID;
    if (!h->red_valid) return fail(h, DQMC_ERR_STATE, "call dqmc_reduce first");
    *out = h->red_stats;
    return DQMC_OK;
}


// CheckerboardTrue with the bond-group factors kept sparse on the device (stack.jl:185-235): `n_mats` factors in ELL
// form (vals / cols [n_mats][n][kmax], 0-based columns, padding val 0), the diagonal exp(-+dtau mu) per block and
// seven factor sequences (order: B, B^-1, B', X B, X B^-1, X eT, eTinv X; entries index the factor list).
int dqmc_set_checkerboard(dqmc_handle *h, int32_t kmax, int32_t n_mats, const double *vals, const int32_t *cols,
                          const double *mu, const double *mu_inv, const int32_t *seqs /* [7][32] */,
                          const int32_t *lens /* [7] */)
{
    ENTER(h);
    if (kmax < 1 || kmax > 64 || n_mats < 1 || !vals || !cols || !mu || !mu_inv || !seqs || !lens)
        return fail(h, DQMC_ERR_INVALID, "dqmc_set_checkerboard: bad arguments");
    const size_t cnt = (size_t)n_mats * h->n * kmax;
    for (size_t i = 0; i < cnt; ++i)
        if (cols[i] < 0 || cols[i] >= h->n) return fail(h, DQMC_ERR_INVALID, "dqmc_set_checkerboard: column index out of range");
    for (int q = 0; q < 7; ++q) {
        if (lens[q] < 0 || lens[q] > 32) return fail(h, DQMC_ERR_INVALID, "dqmc_set_checkerboard: sequence too long");
        for (int i = 0; i < lens[q]; ++i)
            if (seqs[q * 32 + i] < 0 || seqs[q * 32 + i] >= n_mats)
                return fail(h, DQMC_ERR_INVALID, "dqmc_set_checkerboard: factor index out of range");
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(dalloc(h, &h->cb.vals, cnt));
    CHK(dalloc(h, &h->cb.cols, cnt));
    CHK(dalloc(h, &h->cb.mu, (size_t)h->nb * h->n));
    CHK(dalloc(h, &h->cb.mu_inv, (size_t)h->nb * h->n));
    HIPCHK(hipMemcpy(h->cb.vals, vals, cnt * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cb.cols, cols, cnt * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cb.mu, mu, (size_t)h->nb * h->n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cb.mu_inv, mu_inv, (size_t)h->nb * h->n * sizeof(double), hipMemcpyHostToDevice));
    h->cb.kmax = kmax; h->cb.n_mats = n_mats;
    for (int q = 0; q < 7; ++q) {
        h->cb.len[q] = lens[q];
        for (int i = 0; i < lens[q]; ++i) h->cb.seq[q][i] = seqs[q * 32 + i];
    }
    h->cb.on = true;
    return DQMC_OK;
}

// the device error word as it stands (not cleared): 0 unless a bounded wait inside a kernel ran out since the last call that
// reported it (bit 1: sweep elimination hand-off, bit 4: one-launch UDT hand-off)
int dqmc_device_errors(dqmc_handle *h, int32_t *word)
{
    ENTER(h);
    if (!word) return DQMC_ERR_INVALID;
    *word = 0;
    if (!h->qr_ws.errflag) return DQMC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(word, h->qr_ws.errflag, sizeof(int), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
// which call sites of udt_AVX_pivot! take the one-launch pre-pivoted form (bit 0: slice-sequence builds and every other caller,
// bit 1 / 2: first / second factorisation of calculate_greens_AVX!); 0 = the pivoted multi-launch form everywhere
int dqmc_udt_one_launch_sites(dqmc_handle *h, int32_t *mask)
{
    if (!h || !mask) return DQMC_ERR_INVALID;
    *mask = h->qrb_sites;
    return DQMC_OK;
}

// diagnostics: cooperative-QR launches whose hand-offs timed out and were redone by the single-workgroup kernel
int dqmc_qr_fallbacks(dqmc_handle *h, int64_t *count)
{
    ENTER(h);
    if (!count) return DQMC_ERR_INVALID;
    *count = 0;
    if (!h->qr_ws.fb) return DQMC_OK;
    int fb[2] = {0, 0};
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(fb, h->qr_ws.fb, sizeof(fb), hipMemcpyDeviceToHost));
    *count = fb[1];
    return DQMC_OK;
}

int dqmc_timing_enable(dqmc_handle *h, int32_t on)
{
    ENTER(h);
    CHK(timing_drain(h));
    h->timing = on != 0;
    for (int i = 0; i < DQMC_K_COUNT; ++i) { h->fam_ms[i] = 0; h->fam_n[i] = 0; }
    return DQMC_OK;
}
int dqmc_timing_get(dqmc_handle *h, double *ms, int64_t *launches)
{
    ENTER(h);
    CHK(timing_drain(h));
    for (int i = 0; i < DQMC_K_COUNT; ++i) {
        if (ms) ms[i] = h->fam_ms[i];
        if (launches) launches[i] = h->fam_n[i];
    }
    return DQMC_OK;
}

// ---------------------------------------------------------------------------
// stand-alone batched primitives (host in / host out)
// ---------------------------------------------------------------------------
namespace {
struct Scratch {  // a throw-away handle-like context for the primitive entry points
    dqmc_handle h;
    int ok = 0;
};
static int scratch_init(dqmc_handle *h, int device_id, int n, int batch)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, DQMC_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, DQMC_ERR_INVALID, "device_id out of range");
    if (n < 1 || n > 1024 || batch < 1) return fail(nullptr, DQMC_ERR_INVALID, "n must be 1..1024 and batch >= 1");
    refresh_kernel_switches();
    h->p.device_id = device_id;
    h->n = h->N = n;
    h->nb = 1;
    h->W = h->units = batch;
    h->nn = (long)n * n;
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipStreamCreate(&h->stream));
    h->cur = h->stream;
    return 0;
}
static void scratch_free(dqmc_handle *h)
{
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *q : h->allocs) (void)hipFree(q);
    h->allocs.clear();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = nullptr;
}
}  // namespace

#define SCHK(expr)                                   \
    do {                                             \
        int rc_ = (expr);                            \
        if (rc_ != 0) {                              \
            g_create_error = h->err.empty() ? g_create_error : h->err; \
            scratch_free(h);                         \
            return rc_;                              \
        }                                            \
    } while (0)
#define SHIP(expr)                                                       \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            scratch_free(h);                                             \
            return DQMC_ERR_HIP;                                         \
        }                                                                \
    } while (0)

int dqmc_vmul(int32_t device_id, int32_t n, int32_t batch, int32_t ta, int32_t tb, const double *A, const double *B,
              double *C)
{
    dqmc_handle hh; dqmc_handle *h = &hh;
    SCHK(scratch_init(h, device_id, n, batch));
    const size_t un = (size_t)batch * h->nn;
    double *dA, *dB, *dC;
    SCHK(dalloc(h, &dA, un, false)); SCHK(dalloc(h, &dB, un, false)); SCHK(dalloc(h, &dC, un));
    SHIP(hipMemcpy(dA, A, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(dB, B, un * sizeof(double), hipMemcpyHostToDevice));
    SCHK(run_gemm(h, gemm_base(h, U_(h, dA), ta != 0, U_(h, dB), tb != 0, dC)));
    SHIP(hipStreamSynchronize(h->stream));
    SHIP(hipMemcpy(C, dC, un * sizeof(double), hipMemcpyDeviceToHost));
    scratch_free(h);
    return DQMC_OK;
}

static int scratch_udt_bufs(dqmc_handle *h)
{
    const size_t un = (size_t)h->units * h->nn, uv = (size_t)h->units * h->n;
    CHK(dalloc(h, &h->qrV, un)); CHK(dalloc(h, &h->qrW, un)); CHK(dalloc(h, &h->qrS, un));
    CHK(dalloc(h, &h->tau, uv)); CHK(dalloc(h, &h->pivot, uv));
    CHK(dalloc(h, &h->trsm_w, (size_t)h->units * ((h->n + 15) / 16) * 256));
    CHK(alloc_qr_workspace(h));
    return 0;
}

int dqmc_udt_pivot(int32_t device_id, int32_t n, int32_t batch, double *U, double *D, double *T, int64_t *pivot,
                   int32_t apply)
{
    dqmc_handle hh; dqmc_handle *h = &hh;
    SCHK(scratch_init(h, device_id, n, batch));
    const size_t un = (size_t)batch * h->nn, uv = (size_t)batch * n;
    double *dU, *dD, *dT, *dTo;
    SCHK(dalloc(h, &dU, un)); SCHK(dalloc(h, &dD, uv)); SCHK(dalloc(h, &dT, un, false)); SCHK(dalloc(h, &dTo, un));
    SCHK(scratch_udt_bufs(h));
    SHIP(hipMemcpy(dT, T, un * sizeof(double), hipMemcpyHostToDevice));
    SCHK(udt(h, dT, dU, dD, dTo, apply != 0));
    SHIP(hipStreamSynchronize(h->stream));
    SCHK(check_qr_workspace(h));
    SHIP(hipMemcpy(U, dU, un * sizeof(double), hipMemcpyDeviceToHost));
    SHIP(hipMemcpy(D, dD, uv * sizeof(double), hipMemcpyDeviceToHost));
    // (the one-launch form writes T out of place also for Val(false))
    SHIP(hipMemcpy(T, (apply || udt_is_fused(h, 0)) ? dTo : dT, un * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int> piv(uv);
    SHIP(hipMemcpy(piv.data(), h->pivot, uv * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < uv; ++i) pivot[i] = (int64_t)piv[i] + 1;
    scratch_free(h);
    return DQMC_OK;
}

int dqmc_rdivp(int32_t device_id, int32_t n, int32_t batch, double *A, const double *T, const int64_t *pivot)
{
    dqmc_handle hh; dqmc_handle *h = &hh;
    SCHK(scratch_init(h, device_id, n, batch));
    const size_t un = (size_t)batch * h->nn, uv = (size_t)batch * n;
    double *dA, *dT;
    SCHK(dalloc(h, &dA, un, false)); SCHK(dalloc(h, &dT, un, false)); SCHK(dalloc(h, &h->pivot, uv));
    SCHK(dalloc(h, &h->trsm_w, (size_t)h->units * ((h->n + 15) / 16) * 256));
    std::vector<int> piv(uv);
    for (size_t i = 0; i < uv; ++i) {
        if (pivot[i] < 1 || pivot[i] > n) { scratch_free(h); return fail(nullptr, DQMC_ERR_INVALID, "pivot entry out of range"); }
        piv[i] = (int)(pivot[i] - 1);
    }
    SHIP(hipMemcpy(dA, A, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(dT, T, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->pivot, piv.data(), uv * sizeof(int), hipMemcpyHostToDevice));
    SCHK(rdivp(h, dA, dT));
    SHIP(hipStreamSynchronize(h->stream));
    SHIP(hipMemcpy(A, dA, un * sizeof(double), hipMemcpyDeviceToHost));
    scratch_free(h);
    return DQMC_OK;
}

int dqmc_calculate_greens(int32_t device_id, int32_t n, int32_t batch, const double *Ul, const double *Dl,
                          const double *Tl, const double *Ur, const double *Dr, const double *Tr, double *G)
{
    dqmc_handle hh; dqmc_handle *h = &hh;
    SCHK(scratch_init(h, device_id, n, batch));
    const size_t un = (size_t)batch * h->nn, uv = (size_t)batch * n;
    double *dG;
    SCHK(dalloc(h, &h->Ul, un, false)); SCHK(dalloc(h, &h->Ur, un, false)); SCHK(dalloc(h, &h->Tl, un, false));
    SCHK(dalloc(h, &h->Tr, un, false)); SCHK(dalloc(h, &h->Dl, uv, false)); SCHK(dalloc(h, &h->Dr, uv, false));
    SCHK(dalloc(h, &dG, un));
    SCHK(scratch_udt_bufs(h));
    SHIP(hipMemcpy(h->Ul, Ul, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->Ur, Ur, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->Tl, Tl, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->Tr, Tr, un * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->Dl, Dl, uv * sizeof(double), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->Dr, Dr, uv * sizeof(double), hipMemcpyHostToDevice));
    SCHK(calculate_greens(h, dG));
    SHIP(hipStreamSynchronize(h->stream));
    SCHK(check_qr_workspace(h));
    SHIP(hipMemcpy(G, dG, un * sizeof(double), hipMemcpyDeviceToHost));
    scratch_free(h);
    return DQMC_OK;
}

int dqmc_mfma_f64_peak(int32_t device_id, int32_t iters, double *tflops)
{
    dqmc_handle hh; dqmc_handle *h = &hh;
    SCHK(scratch_init(h, device_id, 16, 1));
    double *sink;
    SCHK(dalloc(h, &sink, 16));
    hipDeviceProp_t prop;
    SHIP(hipGetDeviceProperties(&prop, device_id));
    // 8 waves per CU = 2 per SIMD, the occupancy of the GEMM launches at 32 units (DQMC_PROBE_WG_PER_CU: other occupancies)
    const int blocks = prop.multiProcessorCount * (getenv("DQMC_PROBE_WG_PER_CU") ? atoi(getenv("DQMC_PROBE_WG_PER_CU")) : 2);
    hipEvent_t a, b;
    SHIP(hipEventCreate(&a)); SHIP(hipEventCreate(&b));
    SHIP(launch_mfma_peak(iters / 8 + 1, blocks, sink, h->stream));  // warm up
    SHIP(hipEventRecord(a, h->stream));
    SHIP(launch_mfma_peak(iters, blocks, sink, h->stream));
    SHIP(hipEventRecord(b, h->stream));
    SHIP(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    SHIP(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 4.0 /*mfma*/ * 2.0 * 16 * 16 * 4;
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    scratch_free(h);
    return DQMC_OK;
}

}  // extern "C"
