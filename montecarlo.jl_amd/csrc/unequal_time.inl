// unequal_time.inl — unequal-time Green's functions for all walkers of a handle.
// Continuation of engine.cpp (uses its static helpers); host-side orchestration of the same device
// kernels as the sweep: GEMM with fused diagonal scalings, cooperative QR / UDT, TRSM.
//
// Mirrors src/flavors/DQMC/unequal_time_stack.jl: UnequalTimeStack with lazily built forward
// (B_l...B_1), backward (B_{l+1}^T...B_M^T) and inverse (B^-1 per safe_mult block) UDT stacks
// (:54-246), calculate_greens(mc, k, l) = calculate_greens_full1!/full2! (:288-605), GreensIterator
// (:644-715) and CombinedGreensIterator (:746-883).  The reference borrows the DQMCStack
// temporaries Ul..Tr, tmp1, tmp2, curr_U; here the path owns private copies, so the sweep state
// is never disturbed.  All walkers share current_slice, hence one set of stack indices.

struct UTStack {
    int nr = 0;  // length(mc.s.ranges)
    double *fu = nullptr, *ft = nullptr, *fd = nullptr;  // forward  [nr+1]
    double *bu = nullptr, *bt = nullptr, *bd = nullptr;  // backward [nr+1]
    double *iu = nullptr, *it = nullptr, *id = nullptr;  // inverse  [nr]
    std::vector<char> inv_done;
    int forward_idx = 1, backward_idx = 1;
    long long last_update = -1;
    double *U = nullptr, *T = nullptr, *D = nullptr, *greens = nullptr, *tmp = nullptr;  // uts.U/D/T/greens/tmp
    double *Ul = nullptr, *Ur = nullptr, *Tl = nullptr, *Tr = nullptr, *Dl = nullptr, *Dr = nullptr;
    double *s1 = nullptr, *s2 = nullptr, *curr = nullptr;
    double *out[3] = {nullptr, nullptr, nullptr};  // results handed out: G(k,l) in out[0]; (G0l, Gl0, Gll)
    double *g00 = nullptr;                            // greens!(mc) kept for the packed kernels
    double *sus_per_walker = nullptr, *sus_acc = nullptr;
    size_t sus_n = 0;                                 // doubles in sus_acc (last = sample count)
    int it_kind = 0;  // 0 none, 1 GreensIterator, 2 CombinedGreensIterator
    int it_pos = 0, it_l = 0, it_recalc = 0;
};

static void ut_free(dqmc_handle *h)
{
    delete h->ut;  // device buffers are owned by h->allocs
    h->ut = nullptr;
}
static int ut_reset_accumulators(dqmc_handle *h)
{
    if (h->ut && h->ut->sus_acc) HIPCHK(hipMemsetAsync(h->ut->sus_acc, 0, h->ut->sus_n * sizeof(double), h->stream));
    return 0;
}
static double *ut_slot(dqmc_handle *h, double *base, int idx) { return base + (long)idx * h->units * h->nn; }
static double *ut_dslot(dqmc_handle *h, double *base, int idx) { return base + (long)idx * h->units * h->n; }

static int ut_init(dqmc_handle *h)
{
    if (h->ut) return 0;
    UTStack *u = new UTStack();
    h->ut = u;
    u->nr = h->K;
    const size_t mat = (size_t)h->units * h->nn, vec = (size_t)h->units * h->n;
    const size_t m = (size_t)u->nr + 1;
    CHK(dalloc(h, &u->fu, m * mat)); CHK(dalloc(h, &u->ft, m * mat)); CHK(dalloc(h, &u->fd, m * vec));
    CHK(dalloc(h, &u->bu, m * mat)); CHK(dalloc(h, &u->bt, m * mat)); CHK(dalloc(h, &u->bd, m * vec));
    CHK(dalloc(h, &u->iu, (m - 1) * mat)); CHK(dalloc(h, &u->it, (m - 1) * mat)); CHK(dalloc(h, &u->id, (m - 1) * vec));
    double **mats[] = {&u->U, &u->T, &u->greens, &u->tmp, &u->Ul, &u->Ur, &u->Tl, &u->Tr, &u->s1, &u->s2, &u->curr,
                       &u->out[0], &u->out[1], &u->out[2]};
    for (double **p : mats) CHK(dalloc(h, p, mat));
    CHK(dalloc(h, &u->g00, mat));
    CHK(dalloc(h, &u->D, vec)); CHK(dalloc(h, &u->Dl, vec)); CHK(dalloc(h, &u->Dr, vec));
    u->inv_done.assign(u->nr, 0);
    // identities at the ends (unequal_time_stack.jl:90-96)
    CHK(set_identity(h, ut_slot(h, u->fu, 0))); CHK(set_ones(h, ut_dslot(h, u->fd, 0))); CHK(set_identity(h, ut_slot(h, u->ft, 0)));
    CHK(set_identity(h, ut_slot(h, u->bu, u->nr))); CHK(set_ones(h, ut_dslot(h, u->bd, u->nr)));
    CHK(set_identity(h, ut_slot(h, u->bt, u->nr)));
    u->forward_idx = 1;
    u->backward_idx = u->nr + 1;
    u->last_update = -1;
    return 0;
}
// `s.last_update != mc.last_sweep` (:164-169): the stacks belong to an older configuration
static void ut_check_stale(dqmc_handle *h)
{
    UTStack *u = h->ut;
    if (u->last_update != h->conf_version) {
        u->last_update = h->conf_version;
        std::fill(u->inv_done.begin(), u->inv_done.end(), 0);
        u->forward_idx = 1;
        u->backward_idx = u->nr + 1;
    }
}

// ---- small wrappers ---------------------------------------------------------------------------
static int ut_gemm(dqmc_handle *h, MatRef A, int tA, MatRef B, int tB, double *C, VecSrc row = vs_none(),
                   VecSrc col = vs_none(), VecSrc k = vs_none(), int row_first = 0, double alpha = 1.0)
{
    GemmArgs g = gemm_base(h, A, tA, B, tB, C);
    g.rowscale = row; g.colscale = col; g.kscale = k; g.row_first = row_first; g.alpha = alpha;
    return run_gemm(h, g);
}
static int ut_scale(dqmc_handle *h, double *O, const double *A, VecSrc row, VecSrc col, int row_first)
{
    Timed t(h, DQMC_K_MISC);
    HIPCHK(launch_scale_mat(h->n, h->nb, h->units, O, A, h->nn, row, col, row_first, h->stream));
    return 0;
}
enum { UT_LEFT = 0, UT_DAGGER_LEFT = 1, UT_INV_LEFT = 2, UT_INV_RIGHT = 3 };
// one slice-matrix multiply out = op(B_slice) (x) X (slice_matrices.jl:42-76), optional column scaling
// of the product fused into the epilogue
static int ut_slice_mul(dqmc_handle *h, int kind, int slice, const double *X, double *out, VecSrc col = vs_none())
{
    GemmArgs g;
    switch (kind) {
    case UT_LEFT:  // (eT2 eV) X
        g = gemm_base(h, C_(h, h->eT2), 0, U_(h, X), 0, out);
        g.kscale = vs_conf(h, slice, +1);
        break;
    case UT_DAGGER_LEFT:  // (eT2 eV)' X = eV eT2' X
        g = gemm_base(h, C_(h, h->eT2), 1, U_(h, X), 0, out);
        g.rowscale = vs_conf(h, slice, +1);
        g.row_first = 1;
        break;
    case UT_INV_LEFT:  // (eV^-1 eTinv2) X
        g = gemm_base(h, C_(h, h->eTinv2), 0, U_(h, X), 0, out);
        g.rowscale = vs_conf(h, slice, -1);
        g.row_first = 1;
        break;
    default:  // X (eV^-1 eTinv2)
        g = gemm_base(h, U_(h, X), 0, C_(h, h->eTinv2), 0, out);
        g.kscale = vs_conf(h, slice, -1);
        break;
    }
    g.colscale = col;
    return run_gemm(h, g);
}
// in-place form on a named buffer of the stack: *M <- op(B_slice) (x) *M (the buffers swap roles)
static int ut_slice_mul_inplace(dqmc_handle *h, int kind, int slice, double **M, double **spare)
{
    CHK(ut_slice_mul(h, kind, slice, *M, *spare));
    std::swap(*M, *spare);
    return 0;
}
// X <- B_last ... B_first X (step +1) / B'_last ... (step -1) etc. for a run of slices, ending with
// "* Diagonal(D)"; the result lands in *res (one of the two ping-pong buffers)
static int ut_chain(dqmc_handle *h, int kind, int first, int last, int step, const double *X, const double *Dcol,
                    double **res)
{
    UTStack *u = h->ut;
    double *a = u->curr, *b = u->s1;
    const double *src = X;
    const int count = (last - first) / step + 1;
    if ((step > 0 && first > last) || (step < 0 && first < last)) {
        CHK(ut_scale(h, a, X, vs_none(), Dcol ? vs_arr(Dcol, h->n) : vs_none(), 0));
        *res = a;
        return 0;
    }
    int i = 0;
    for (int sl = first; step > 0 ? sl <= last : sl >= last; sl += step, ++i) {
        const bool fin = i == count - 1;
        CHK(ut_slice_mul(h, kind, sl, src, a, fin && Dcol ? vs_arr(Dcol, h->n) : vs_none()));
        src = a;
        std::swap(a, b);
    }
    *res = const_cast<double *>(src);
    return 0;
}
// udt_AVX_pivot!(U, D, input) with the pivot applied: A is consumed, T goes to Tout
static int ut_udt(dqmc_handle *h, double *A, double *Uout, double *Dout, double *Tout) { return udt(h, A, Uout, Dout, Tout, 1); }

// ---- lazy stack builds (:162-246) -----------------------------------------------------------------
static int ut_lazy_build_forward(dqmc_handle *h, int upto)
{
    UTStack *u = h->ut;
    ut_check_stale(h);
    for (int idx = u->forward_idx; idx <= upto - 1; ++idx) {  // 1-based as in the reference
        double *res;
        CHK(ut_chain(h, UT_LEFT, (idx - 1) * h->s + 1, idx * h->s, +1, ut_slot(h, u->fu, idx - 1),
                     ut_dslot(h, u->fd, idx - 1), &res));
        CHK(ut_udt(h, res, ut_slot(h, u->fu, idx), ut_dslot(h, u->fd, idx), u->s2));
        CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, ut_slot(h, u->ft, idx - 1)), 0, ut_slot(h, u->ft, idx)));
    }
    u->forward_idx = std::max(upto, u->forward_idx);
    return 0;
}
static int ut_lazy_build_backward(dqmc_handle *h, int downto)
{
    UTStack *u = h->ut;
    ut_check_stale(h);
    for (int idx = u->backward_idx - 1; idx >= downto; --idx) {
        double *res;
        CHK(ut_chain(h, UT_DAGGER_LEFT, idx * h->s, (idx - 1) * h->s + 1, -1, ut_slot(h, u->bu, idx),
                     ut_dslot(h, u->bd, idx), &res));
        CHK(ut_udt(h, res, ut_slot(h, u->bu, idx - 1), ut_dslot(h, u->bd, idx - 1), u->s2));
        CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, ut_slot(h, u->bt, idx)), 0, ut_slot(h, u->bt, idx - 1)));
    }
    u->backward_idx = std::min(downto, u->backward_idx);
    return 0;
}
static int ut_lazy_build_inv(dqmc_handle *h, int from, int to)
{
    UTStack *u = h->ut;
    ut_check_stale(h);
    for (int idx = from; idx <= to; ++idx) {
        if (u->inv_done[idx - 1]) continue;
        u->inv_done[idx - 1] = 1;
        CHK(set_identity(h, u->s2));
        double *res;
        CHK(ut_chain(h, UT_INV_LEFT, idx * h->s, (idx - 1) * h->s + 1, -1, u->s2, nullptr, &res));
        CHK(ut_udt(h, res, ut_slot(h, u->iu, idx - 1), ut_dslot(h, u->id, idx - 1), ut_slot(h, u->it, idx - 1)));
    }
    return 0;
}
// build_stack(mc, s::UnequalTimeStack) (:106-160)
static int ut_build_stack(dqmc_handle *h)
{
    UTStack *u = h->ut;
    CHK(ut_lazy_build_forward(h, u->nr + 1));
    CHK(ut_lazy_build_backward(h, 1));
    CHK(ut_lazy_build_inv(h, 1, u->nr));
    return 0;
}

// ---- UDT blocks (:322-443) ----------------------------------------------------------------------------
// uts.U uts.D uts.T = B_{low+1}^-1 ... B_high^-1
static int ut_compute_inverse_udt_block(dqmc_handle *h, int low, int high)
{
    UTStack *u = h->ut;
    const int s = h->s, n = h->n;
    const int lower = (low + 1 + s - 2) / s + 1, upper = high / s;
    CHK(ut_lazy_build_inv(h, lower, upper));
    CHK(set_identity(h, u->U)); CHK(set_ones(h, u->D)); CHK(set_identity(h, u->T));
    for (int idx = lower; idx <= upper; ++idx) {
        // tmp1 = (Diagonal(D) (T inv_u)) Diagonal(inv_d)
        CHK(ut_gemm(h, U_(h, u->T), 0, U_(h, ut_slot(h, u->iu, idx - 1)), 0, u->s1, vs_arr(u->D, n),
                    vs_arr(ut_dslot(h, u->id, idx - 1), n), vs_none(), 1));
        CHK(ut_udt(h, u->s1, u->s2, u->D, u->tmp));                                            // tmp2, D, tmp1
        CHK(ut_gemm(h, U_(h, u->tmp), 0, U_(h, ut_slot(h, u->it, idx - 1)), 0, u->T));       // T = tmp1 inv_t
        CHK(ut_gemm(h, U_(h, u->U), 0, U_(h, u->s2), 0, u->s1));                               // U = U tmp2
        std::swap(u->U, u->s1);
    }
    const int lower_slice = (lower - 1) * s + 1, upper_slice = upper * s;
    const int top = std::min(lower_slice - 1, high);
    if (top >= low + 1) {
        double *res;
        CHK(ut_chain(h, UT_INV_LEFT, top, low + 1, -1, u->U, u->D, &res));  // ... then tmp1 = U Diagonal(D)
        CHK(ut_udt(h, res, u->U, u->D, u->s2));
        CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, u->T), 0, u->tmp));
        std::swap(u->T, u->tmp);
    }
    for (int sl = std::max(upper_slice + 1, top + 1); sl <= high; ++sl)
        CHK(ut_slice_mul_inplace(h, UT_INV_RIGHT, sl, &u->T, &u->tmp));
    return 0;
}
// Ul Dl Tl = B_slice ... B_1
static int ut_compute_forward_udt_block(dqmc_handle *h, int slice)
{
    UTStack *u = h->ut;
    const int s = h->s;
    const int idx = slice >= 1 ? (slice - 1) / s : 0;
    CHK(ut_lazy_build_forward(h, idx + 1));
    double *res;
    CHK(ut_chain(h, UT_LEFT, s * idx + 1, slice, +1, ut_slot(h, u->fu, idx), ut_dslot(h, u->fd, idx), &res));
    CHK(ut_udt(h, res, u->Ul, u->Dl, u->s2));
    CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, ut_slot(h, u->ft, idx)), 0, u->Tl));
    return 0;
}
// (Ur Dr Tr)' = B_M ... B_{slice+1}
static int ut_compute_backward_udt_block(dqmc_handle *h, int slice)
{
    UTStack *u = h->ut;
    const int s = h->s;
    const int idx = (slice + s - 1) / s;
    CHK(ut_lazy_build_backward(h, idx + 1));
    double *res;
    CHK(ut_chain(h, UT_DAGGER_LEFT, s * idx, slice + 1, -1, ut_slot(h, u->bu, idx), ut_dslot(h, u->bd, idx), &res));
    CHK(ut_udt(h, res, u->Ur, u->Dr, u->s2));
    CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, ut_slot(h, u->bt, idx)), 0, u->Tr));
    return 0;
}

// ---- calculate_greens_full1! (:447-530), slice1 >= slice2 ---------------------------------------------
static int ut_full1(dqmc_handle *h, int slice1, int slice2)
{
    UTStack *u = h->ut;
    const int n = h->n;
    CHK(ut_compute_inverse_udt_block(h, slice2, slice1));
    CHK(ut_compute_forward_udt_block(h, slice2));
    CHK(ut_compute_backward_udt_block(h, slice1));
    // B1: greens = Diagonal(Dl) ((Tl Tr') Diagonal(Dr))
    CHK(ut_gemm(h, U_(h, u->Tl), 0, U_(h, u->Tr), 1, u->greens, vs_arr(u->Dl, n), vs_arr(u->Dr, n), vs_none(), 0));
    CHK(udt(h, u->greens, u->Tr, u->Dr, nullptr, 0));                       // Tr, Dr, greens (pivot kept)
    CHK(ut_gemm(h, U_(h, u->Ul), 0, U_(h, u->Tr), 0, u->s1));               // B2: Tl = Ul Tr
    std::swap(u->Tl, u->s1);
    CHK(rdivp(h, u->Ur, u->greens));                                        //     Ur = Ur / greens
    // B3: Tr = Diagonal(1/max(1,D)) (U' Tl) Diagonal(min(1,Dr))
    CHK(ut_gemm(h, U_(h, u->U), 1, U_(h, u->Tl), 0, u->Tr, vs_maxinv(u->D, n), vs_min1(u->Dr, n), vs_none(), 1));
    // B4: Tl = Diagonal(min(1,D)) (T Ur) Diagonal(1/max(1,Dr))
    CHK(ut_gemm(h, U_(h, u->T), 0, U_(h, u->Ur), 0, u->s1, vs_min1(u->D, n), vs_maxinv(u->Dr, n), vs_none(), 1));
    std::swap(u->Tl, u->s1);
    {   // sum, UDT
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_mat_add(u->Tl, u->Tr, (size_t)h->units * h->nn, h->stream));
    }
    CHK(udt(h, u->Tl, u->Tr, u->Dl, nullptr, 0));                           // Tr, Dl, Tl
    {   // B5: Dr = 1/max(1,Dr); Ul = Diagonal(Dr) / Tl
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_vec_map(n, h->nb, h->units, u->Dr, n, vs_maxinv(u->Dr, n), h->stream));
        HIPCHK(launch_set_diag(n, h->nb, h->units, u->Ul, h->nn, vs_arr(u->Dr, n), h->stream));
    }
    CHK(rdivp(h, u->Ul, u->Tl));
    //     greens = ((Ul Diagonal(1/Dl)) Tr') Diagonal(1/max(1,D))
    CHK(ut_gemm(h, U_(h, u->Ul), 0, U_(h, u->Tr), 1, u->s1, vs_none(), vs_maxinv(u->D, n), vs_inv(u->Dl, n)));
    CHK(ut_gemm(h, U_(h, u->s1), 0, U_(h, u->U), 1, u->s2));                // B6: Tr = greens U'
    CHK(ut_gemm(h, U_(h, u->Ur), 0, U_(h, u->s2), 0, u->greens));           //     greens = Ur Tr
    return 0;
}

// ---- calculate_greens_full2! (:534-605), slice1 <= slice2 ----------------------------------------------
static int ut_full2(dqmc_handle *h, int slice1, int slice2)
{
    UTStack *u = h->ut;
    const int n = h->n;
    CHK(ut_compute_inverse_udt_block(h, slice1, slice2));
    CHK(ut_compute_forward_udt_block(h, slice1));
    CHK(ut_compute_backward_udt_block(h, slice2));
    // B1: greens = (Diagonal(Dl) (Tl Tr')) Diagonal(Dr)
    CHK(ut_gemm(h, U_(h, u->Tl), 0, U_(h, u->Tr), 1, u->greens, vs_arr(u->Dl, n), vs_arr(u->Dr, n), vs_none(), 1));
    CHK(udt(h, u->greens, u->Tr, u->Dr, nullptr, 0));                       // Tr, Dr, greens
    CHK(ut_gemm(h, U_(h, u->Ul), 0, U_(h, u->Tr), 0, u->s1));               // B2: Tl = Ul Tr
    std::swap(u->Tl, u->s1);
    //     Ul = Diagonal(1/max(1,D)) (U' Tl) Diagonal(min(1,Dr))
    CHK(ut_gemm(h, U_(h, u->U), 1, U_(h, u->Tl), 0, u->Ul, vs_maxinv(u->D, n), vs_min1(u->Dr, n), vs_none(), 1));
    CHK(ut_gemm(h, U_(h, u->T), 0, U_(h, u->Ur), 0, u->s1));                // B3: s.U = T Ur
    CHK(rdivp(h, u->s1, u->greens));                                        //     s.U = s.U / greens
    //     Tr = (Diagonal(min(1,D)) s.U) Diagonal(1/max(1,Dr))
    CHK(ut_scale(h, u->Tr, u->s1, vs_min1(u->D, n), vs_maxinv(u->Dr, n), 1));
    {   // sum, udt
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_mat_add(u->Tr, u->Ul, (size_t)h->units * h->nn, h->stream));
    }
    CHK(udt(h, u->Tr, u->Ul, u->Dl, nullptr, 0));                           // Ul, Dl, Tr
    {   // B4: s.U = Diagonal(min(1,Dr)) / Tr
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_set_diag(n, h->nb, h->units, u->s1, h->nn, vs_min1(u->Dr, n), h->stream));
    }
    CHK(rdivp(h, u->s1, u->Tr));
    //     Ur = ((s.U Diagonal(1/Dl)) Ul') Diagonal(min(1,D))
    CHK(ut_gemm(h, U_(h, u->s1), 0, U_(h, u->Ul), 1, u->s2, vs_none(), vs_min1(u->D, n), vs_inv(u->Dl, n)));
    CHK(ut_gemm(h, U_(h, u->s2), 0, U_(h, u->T), 0, u->s1));                // B6: Tr = Ur T
    CHK(ut_gemm(h, U_(h, u->Tl), 0, U_(h, u->s1), 0, u->greens, vs_none(), vs_none(), vs_none(), 0, -1.0));
    return 0;
}
// calculate_greens(mc, slice1, slice2) (:288-303) into uts.greens
static int ut_calculate_greens(dqmc_handle *h, int slice1, int slice2)
{
    return slice1 >= slice2 ? ut_full1(h, slice1, slice2) : ut_full2(h, slice1, slice2);
}
// _greens!(mc, target, source, temp) (DQMC.jl:721-730): eThalfplus (source eThalfminus)
static int ut_true(dqmc_handle *h, const double *src, double *dst)
{
    UTStack *u = h->ut;
    double *t = (src == u->s2 || dst == u->s2) ? u->curr : u->s2;
    CHK(ut_gemm(h, U_(h, src), 0, C_(h, h->eT), 0, t));
    CHK(ut_gemm(h, C_(h, h->eTinv), 0, U_(h, t), 0, dst));
    return 0;
}

// ---- GreensIterator{:, l} (:644-715) --------------------------------------------------------------------
static int ut_gi_refactor(dqmc_handle *h)  // copyto!(s.T, s.greens); udt_AVX_pivot!(s.U, s.D, s.T)
{
    UTStack *u = h->ut;
    CHK(copy_mat(h, u->s1, u->greens));
    return ut_udt(h, u->s1, u->U, u->D, u->T);
}
static int ut_gi_begin(dqmc_handle *h, int l, int recalc)
{
    UTStack *u = h->ut;
    CHK(ut_full1(h, l, l));
    CHK(ut_true(h, u->greens, u->out[0]));
    CHK(ut_gi_refactor(h));
    u->it_kind = 1; u->it_l = l; u->it_pos = l + 1; u->it_recalc = recalc;
    return 0;
}
static int ut_gi_next(dqmc_handle *h, int *k_out)
{
    UTStack *u = h->ut;
    const int k = u->it_pos, n = h->n;
    if (k > h->M) { *k_out = -1; u->it_kind = 0; return 0; }
    if (k % u->it_recalc == 0) {
        CHK(ut_full1(h, k, u->it_l));
        CHK(ut_true(h, u->greens, u->out[0]));
        CHK(ut_gi_refactor(h));
    } else if (k % h->s == 0) {
        CHK(ut_slice_mul_inplace(h, UT_LEFT, k, &u->U, &u->s1));
        CHK(ut_gemm(h, U_(h, u->U), 0, U_(h, u->T), 0, u->tmp, vs_none(), vs_none(), vs_arr(u->D, n)));  // (U D) T
        CHK(ut_scale(h, u->curr, u->U, vs_none(), vs_arr(u->D, n), 0));
        CHK(ut_udt(h, u->curr, u->U, u->D, u->s1));
        CHK(ut_gemm(h, U_(h, u->s1), 0, U_(h, u->T), 0, u->s2));
        std::swap(u->T, u->s2);
        CHK(ut_true(h, u->tmp, u->out[0]));
    } else {
        CHK(ut_slice_mul_inplace(h, UT_LEFT, k, &u->U, &u->s1));
        CHK(ut_gemm(h, U_(h, u->U), 0, U_(h, u->T), 0, u->tmp, vs_none(), vs_none(), vs_arr(u->D, n)));
        CHK(ut_true(h, u->tmp, u->out[0]));
    }
    *k_out = k;
    u->it_pos = k + 1;
    return 0;
}

// ---- CombinedGreensIterator (:746-883): (G0l, Gl0, Gll) for l = 1..slices ----------------------------------
static int ut_cgi_begin(dqmc_handle *h, int recalc)
{
    UTStack *u = h->ut;
    if (h->current_slice != 1) return fail(h, DQMC_ERR_STATE, "CombinedGreensIterator needs current_slice == 1");
    CHK(ut_build_stack(h));
    CHK(copy_mat(h, u->s1, h->greens));
    CHK(ut_udt(h, u->s1, u->Ul, u->Dl, u->Tl));
    CHK(copy_mat(h, u->U, u->Ul)); CHK(copy_vec(h, u->D, u->Dl)); CHK(copy_mat(h, u->T, u->Tl));
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_sub_identity(h->n, h->units, u->s1, h->greens, h->nn, h->stream));
    }
    CHK(ut_udt(h, u->s1, u->Ur, u->Dr, u->Tr));
    u->it_kind = 2; u->it_pos = 1; u->it_recalc = recalc;
    return 0;
}
static int ut_cgi_next(dqmc_handle *h, int *l_out)
{
    UTStack *u = h->ut;
    const int l = u->it_pos, n = h->n;
    if (l > h->M) { *l_out = -1; u->it_kind = 0; return 0; }
    if (l % u->it_recalc == 0) {
        // every buffer of the iterator state is a temporary of full1/full2: results are parked in out[]
        CHK(ut_full1(h, l, 0));
        CHK(copy_mat(h, u->out[1], u->greens));       // Gl0 (effective)
        CHK(ut_full2(h, 0, l));
        CHK(copy_mat(h, u->out[0], u->greens));       // G0l (effective)
        CHK(ut_full1(h, l, l));
        CHK(ut_true(h, u->greens, u->out[2]));        // Gll
        CHK(copy_mat(h, u->s1, u->greens));
        CHK(ut_udt(h, u->s1, u->U, u->D, u->T));
        CHK(copy_mat(h, u->s1, u->out[1]));
        CHK(ut_true(h, u->s1, u->out[1]));
        CHK(ut_udt(h, u->s1, u->Ul, u->Dl, u->Tl));
        CHK(copy_mat(h, u->s1, u->out[0]));
        CHK(ut_true(h, u->s1, u->out[0]));
        CHK(ut_udt(h, u->s1, u->Ur, u->Dr, u->Tr));
    } else {
        CHK(ut_slice_mul_inplace(h, UT_LEFT, l, &u->Ul, &u->s1));
        CHK(ut_slice_mul_inplace(h, UT_INV_RIGHT, l, &u->Tr, &u->s1));
        CHK(ut_slice_mul_inplace(h, UT_LEFT, l, &u->U, &u->s1));
        CHK(ut_slice_mul_inplace(h, UT_INV_RIGHT, l, &u->T, &u->s1));
        // effective Green's functions of this step: (Ul Dl) Tl, (Ur Dr) Tr resp. Ur (Dr Tr), (U D) T
        CHK(ut_gemm(h, U_(h, u->Ul), 0, U_(h, u->Tl), 0, u->tmp, vs_none(), vs_none(), vs_arr(u->Dl, n)));
        CHK(ut_true(h, u->tmp, u->out[1]));
        CHK(ut_gemm(h, U_(h, u->Ur), 0, U_(h, u->Tr), 0, u->tmp, vs_none(), vs_none(), vs_arr(u->Dr, n)));
        CHK(ut_true(h, u->tmp, u->out[0]));
        CHK(ut_gemm(h, U_(h, u->U), 0, U_(h, u->T), 0, u->tmp, vs_none(), vs_none(), vs_arr(u->D, n)));
        CHK(ut_true(h, u->tmp, u->out[2]));
        if (l % h->s == 0) {  // stabilization (:820-858)
            // Gl0: Ul, Dl, tmp1 = udt(Ul Dl); Tl = tmp1 Tl
            CHK(ut_scale(h, u->curr, u->Ul, vs_none(), vs_arr(u->Dl, n), 0));
            CHK(ut_udt(h, u->curr, u->Ul, u->Dl, u->s1));
            CHK(ut_gemm(h, U_(h, u->s1), 0, U_(h, u->Tl), 0, u->tmp));
            std::swap(u->Tl, u->tmp);
            // G0l: tmp2, Dr, Tr = udt(Dr Tr); Ur = Ur tmp2
            CHK(ut_scale(h, u->curr, u->Tr, vs_arr(u->Dr, n), vs_none(), 1));
            CHK(ut_udt(h, u->curr, u->s1, u->Dr, u->Tr));
            CHK(ut_gemm(h, U_(h, u->Ur), 0, U_(h, u->s1), 0, u->tmp));
            std::swap(u->Ur, u->tmp);
            // Gll: curr_U, D, tmp = udt(U D); U = tmp T; T = D U; tmp, D, T = udt(T); U = curr_U tmp
            CHK(ut_scale(h, u->curr, u->U, vs_none(), vs_arr(u->D, n), 0));
            CHK(ut_udt(h, u->curr, u->s2, u->D, u->s1));                     // s2 = curr_U
            CHK(ut_gemm(h, U_(h, u->s1), 0, U_(h, u->T), 0, u->tmp, vs_arr(u->D, n)));  // Diagonal(D) (tmp T)
            CHK(copy_mat(h, u->U, u->s2));                                    // keep curr_U (udt uses s2? no: park in U)
            CHK(ut_udt(h, u->tmp, u->s1, u->D, u->T));                        // s1 = tmp (U''), D, T
            CHK(ut_gemm(h, U_(h, u->U), 0, U_(h, u->s1), 0, u->tmp));
            std::swap(u->U, u->tmp);
        }
    }
    *l_out = l;
    u->it_pos = l + 1;
    return 0;
}

// ---- C ABI ------------------------------------------------------------------------------------------------
#define NEED_UT(h)            \
    NEED_PREPARED(h);         \
    CHK(ut_init(h))
static int ut_copy_out(dqmc_handle *h, const double *src, int32_t w, double *out)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, src + (size_t)w * h->nb * h->nn, sizeof(double) * h->nb * h->nn, hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_ut_build_stack(dqmc_handle *h)
{
    ENTER(h); NEED_UT(h);
    CHK(ut_build_stack(h));
    return dqmc_synchronize(h);
}
int dqmc_ut_get_stack(dqmc_handle *h, int32_t w, int32_t which, int32_t idx, double *U, double *D, double *T)
{
    ENTER(h); WALKER_OK(h, w); NEED_UT(h);
    UTStack *u = h->ut;
    const int len = which == 2 ? u->nr : u->nr + 1;
    if (which < 0 || which > 2 || idx < 0 || idx >= len) return fail(h, DQMC_ERR_INVALID, "bad stack selector");
    double *ub = which == 0 ? u->fu : which == 1 ? u->bu : u->iu;
    double *tb = which == 0 ? u->ft : which == 1 ? u->bt : u->it;
    double *db = which == 0 ? u->fd : which == 1 ? u->bd : u->id;
    CHK(ut_copy_out(h, ut_slot(h, ub, idx), w, U));
    CHK(ut_copy_out(h, ut_slot(h, tb, idx), w, T));
    HIPCHK(hipMemcpy(D, ut_dslot(h, db, idx) + (size_t)w * h->nb * h->n, sizeof(double) * h->nb * h->n,
                     hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_ut_greens(dqmc_handle *h, int32_t slice1, int32_t slice2, int32_t effective)
{
    ENTER(h); NEED_UT(h);
    if (slice1 < 0 || slice1 > h->M || slice2 < 0 || slice2 > h->M)
        return fail(h, DQMC_ERR_INVALID, "slice out of range 0..slices");
    UTStack *u = h->ut;
    u->it_kind = 0;  // greens!(mc, k, l) breaks a running iteration (unequal_time_stack.jl:664-667)
    CHK(ut_calculate_greens(h, slice1, slice2));
    if (effective) CHK(copy_mat(h, u->out[0], u->greens));
    else CHK(ut_true(h, u->greens, u->out[0]));
    return dqmc_synchronize(h);
}
int dqmc_ut_get(dqmc_handle *h, int32_t w, int32_t which, double *out)
{
    ENTER(h); WALKER_OK(h, w); NEED_UT(h);
    if (which < 0 || which > 2) return fail(h, DQMC_ERR_INVALID, "which must be 0, 1 or 2");
    return ut_copy_out(h, h->ut->out[which], w, out);
}
int dqmc_ut_export(dqmc_handle *h, int32_t which, void *device_out)
{
    ENTER(h); NEED_UT(h);
    if (which < 0 || which > 2) return fail(h, DQMC_ERR_INVALID, "which must be 0, 1 or 2");
    HIPCHK(hipMemcpyAsync(device_out, h->ut->out[which], sizeof(double) * h->units * h->nn, hipMemcpyDeviceToDevice,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
int dqmc_greens_iterator_begin(dqmc_handle *h, int32_t l, int32_t recalculate)
{
    ENTER(h); NEED_UT(h);
    if (l < 0 || l > h->M || recalculate < 1) return fail(h, DQMC_ERR_INVALID, "bad GreensIterator arguments");
    CHK(ut_gi_begin(h, l, recalculate));
    return dqmc_synchronize(h);
}
int dqmc_greens_iterator_next(dqmc_handle *h, int32_t *k)
{
    ENTER(h); NEED_UT(h);
    if (!k) return DQMC_ERR_INVALID;
    if (h->ut->it_kind != 1) return fail(h, DQMC_ERR_STATE, "no GreensIterator in progress");
    int kk = -1;
    CHK(ut_gi_next(h, &kk));
    *k = kk;
    return dqmc_synchronize(h);
}
int dqmc_combined_iterator_begin(dqmc_handle *h, int32_t recalculate)
{
    ENTER(h); NEED_UT(h);
    if (recalculate < 1) return fail(h, DQMC_ERR_INVALID, "bad CombinedGreensIterator arguments");
    CHK(ut_cgi_begin(h, recalculate));
    return dqmc_synchronize(h);
}
int dqmc_combined_iterator_next(dqmc_handle *h, int32_t *l)
{
    ENTER(h); NEED_UT(h);
    if (!l) return DQMC_ERR_INVALID;
    if (h->ut->it_kind != 2) return fail(h, DQMC_ERR_STATE, "no CombinedGreensIterator in progress");
    int ll = -1;
    CHK(ut_cgi_next(h, &ll));
    *l = ll;
    return dqmc_synchronize(h);
}

// ---- susceptibilities: apply!(::CombinedGreensIterator, ...) (generic.jl:226-243) on the device -------------
// charge_density_susceptibility, spin_density_susceptibility(:x/:y/:z), pairing_susceptibility
// (measurements.jl:57-58,142-144,207): sum over l of kernel(G00, G0l, Gl0, Gll), finish! * delta_tau / N
static int ut_sus_layout(dqmc_handle *h)
{
    UTStack *u = h->ut;
    const size_t want = 4 * (size_t)h->n_dirs + (size_t)h->n_dirs * h->K_loc * h->K_loc + 1;
    if (u->sus_n == want) return 0;
    u->sus_n = want;
    h->red_valid = false;  // (re)sized: the last reduction is void
    CHK(dalloc(h, &u->sus_per_walker, (size_t)h->W * (want - 1)));
    CHK(dalloc(h, &u->sus_acc, want));
    return 0;
}
int dqmc_accumulate_susceptibilities(dqmc_handle *h, int32_t recalculate)
{
    ENTER(h); NEED_UT(h);
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    if (recalculate < 1) return fail(h, DQMC_ERR_INVALID, "recalculate must be positive");
    UTStack *u = h->ut;
    CHK(ut_sus_layout(h));
    const long total = (long)u->sus_n - 1;
    CHK(true_greens(h, h->greens));                                 // G00 = greens!(mc)
    CHK(copy_mat(h, u->g00, h->tmp2));
    HIPCHK(hipMemsetAsync(u->sus_per_walker, 0, sizeof(double) * h->W * total, h->stream));  // prepare!
    CHK(ut_cgi_begin(h, recalculate));
    for (;;) {
        int l = -1;
        CHK(ut_cgi_next(h, &l));
        if (l < 0) break;
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_sus_slice(h->n, h->nb, h->p.model_kind, h->W, u->g00, u->out[0], u->out[1], u->out[2], h->nn,
                                h->dir_ptr, h->pair_src, h->pair_trg, h->n_dirs, h->K_loc, h->trg_of, u->sus_per_walker,
                                total, h->stream));
    }
    {
        Timed t(h, DQMC_K_MISC);
        HIPCHK(launch_sus_reduce(h->W, total, h->p.delta_tau, u->sus_per_walker, u->sus_acc, h->stream));
    }
    return dqmc_synchronize(h);
}
int dqmc_susceptibilities_size(dqmc_handle *h, size_t *n)
{
    ENTER(h); NEED_UT(h);
    if (!n) return DQMC_ERR_INVALID;
    if (!h->n_dirs) return fail(h, DQMC_ERR_STATE, "call dqmc_set_pair_directions first");
    CHK(ut_sus_layout(h));
    *n = h->ut->sus_n;
    return DQMC_OK;
}
int dqmc_get_susceptibilities(dqmc_handle *h, double *host_out)
{
    ENTER(h); NEED_UT(h);
    if (!h->ut->sus_acc) return fail(h, DQMC_ERR_STATE, "nothing accumulated");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, h->ut->sus_acc, h->ut->sus_n * sizeof(double), hipMemcpyDeviceToHost));
    return DQMC_OK;
}
int dqmc_export_susceptibilities(dqmc_handle *h, void *device_out)
{
    ENTER(h); NEED_UT(h);
    if (!h->ut->sus_acc) return fail(h, DQMC_ERR_STATE, "nothing accumulated");
    HIPCHK(hipMemcpyAsync(device_out, h->ut->sus_acc, h->ut->sus_n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DQMC_OK;
}
