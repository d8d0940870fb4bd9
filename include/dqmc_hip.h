/*
 * dqmc_hip.h — C ABI of libdqmc_hip.so, the MI355X (gfx950) DQMC sweep engine.
 *
 * The reference (ffreyer/MonteCarlo.jl) has no FFI: its plugin surface is Julia
 * dispatch on `Stack <: AbstractDQMCStack` (src/flavors/DQMC/DQMC.jl:133-136,
 * stack.jl:108,242).  This ABI is what a `HIPDQMCStack <: AbstractDQMCStack`
 * would `ccall` (binding text in INTEGRATION.md).  Every entry point names the
 * reference function/state it stands in for.  All paths are relative to the
 * reference repository root.
 *
 * Conventions
 *  - every function returns 0 on success or a negative dqmc_status; the message
 *    is available from dqmc_last_error().  No C++ exception crosses the ABI.
 *  - matrices are IEEE fp64, column-major, exactly as Julia `Matrix{Float64}`;
 *    the HS field is `Array{Int8,2}` (n_sites x slices, column-major, values ±1);
 *    pivots are Int64 and 1-based as in Julia.
 *  - the caller owns every host buffer; the library owns all device memory.
 *  - numerical events (propagation instability, negative determinant ratio) are
 *    counters, not errors, as in the reference (DQMC.jl:4-47).
 *  - a handle drives one device from one host thread.
 *  - "unit" = (walker, block): attractive model 1 block per walker, repulsive
 *    model 2 blocks (spin up / down, src/linalg/blockdiagonal.jl:13-36).  Buffers
 *    documented as "per walker" hold n_blocks consecutive n x n matrices.
 */
#ifndef DQMC_HIP_H
#define DQMC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dqmc_handle dqmc_handle;

typedef enum {
    DQMC_OK = 0,
    DQMC_ERR_INVALID = -1,  /* bad argument (Julia: error()/@assert, e.g. stack.jl:173) */
    DQMC_ERR_HIP = -2,      /* HIP runtime error */
    DQMC_ERR_NO_DEVICE = -3,
    DQMC_ERR_STATE = -4,    /* call order violated (e.g. sweep before prepare) */
    DQMC_ERR_RNG = -5       /* host-supplied uniform stream exhausted */
} dqmc_status;

enum { DQMC_ATTRACTIVE = 0, DQMC_REPULSIVE = 1 };

/* DQMCParameters (DQMC.jl:52-125) + the model fields the hot path reads
 * (HubbardModelAttractive.jl:26-39, HubbardModelRepulsive.jl:24-41) + the
 * host-computed hopping exponentials (stack.jl:167-181). */
typedef struct {
    int32_t n_sites;                 /* length(lattice) */
    int32_t model_kind;              /* DQMC_ATTRACTIVE / DQMC_REPULSIVE */
    int32_t slices;                  /* mc.p.slices, must be divisible by safe_mult (stack.jl:115) */
    int32_t safe_mult;               /* mc.p.safe_mult */
    int32_t n_walkers;               /* independent Markov chains batched on this device */
    int32_t device_id;
    int32_t check_propagation_error; /* mc.p.check_propagation_error */
    int32_t check_sign_problem;      /* mc.p.check_sign_problem */
    double delta_tau;                /* mc.p.delta_tau */
    double U;                        /* model.U (>= 0, sign carried by model_kind) */
    /* n_blocks consecutive n x n col-major matrices each; copied at create */
    const double *eT;                /* hopping_matrix_exp             = exp(-dtau/2 T) */
    const double *eTinv;             /* hopping_matrix_exp_inv         = exp(+dtau/2 T) */
    const double *eT2;               /* hopping_matrix_exp_squared     */
    const double *eTinv2;            /* hopping_matrix_exp_inv_squared */
} dqmc_params;

/* MagnitudeStats (DQMC.jl:4-31): log10 magnitudes */
typedef struct {
    double max, min, sum;
    int64_t count;
} dqmc_magstats;

/* DQMCAnalysis (DQMC.jl:36-47) for one walker */
typedef struct {
    int64_t prop_local, acc_local;
    dqmc_magstats imaginary_probability; /* always empty: the Hubbard models are real (HubbardModel.jl:52) */
    dqmc_magstats negative_probability;
    dqmc_magstats propagation_error;
} dqmc_stats;

/* ---- lifetime ---------------------------------------------------------- */
/* DQMC(model; ...) + init!(mc) + initialize_stack (DQMC.jl:250-289,337-341; stack.jl:108-159) */
int dqmc_create(const dqmc_params *p, dqmc_handle **out);
int dqmc_destroy(dqmc_handle *h);
/* message of the last failing call on this handle (h may be NULL: create errors) */
const char *dqmc_last_error(const dqmc_handle *h);
/* number of visible HIP devices (0 if none); never fails */
int dqmc_device_count(void);

/* ---- state ------------------------------------------------------------- */
/* mc.conf (HubbardModel.jl:4-5,46-48); conf is n_sites x slices Int8 */
int dqmc_set_conf(dqmc_handle *h, int32_t walker, const int8_t *conf);
int dqmc_get_conf(dqmc_handle *h, int32_t walker, int8_t *conf);
/* compress(mc, model, conf) = BitArray(conf .== 1) and decompress = 2c .- 1 (HubbardModel.jl:56-59,
 * used by ConfigRecorder, src/configurations.jl:24-43): the chunks of Julia's BitArray, element i
 * (1-based column-major) in bit (i-1)%64 of chunk (i-1)/64; ceil(n_sites*slices/64) chunks */
int dqmc_get_conf_bits(dqmc_handle *h, int32_t walker, uint64_t *chunks);
int dqmc_set_conf_bits(dqmc_handle *h, int32_t walker, const uint64_t *chunks);
/* RNG feeding `rand() < p` (DQMC.jl:573).  Test mode: a host-supplied uniform
 * stream consumed with the reference's conditional rule (only when p <= 1). */
int dqmc_set_uniforms(dqmc_handle *h, int32_t walker, const double *u, size_t n);
/* Production mode: Philox4x32-10 counter stream, counter = draw index, key = seed */
int dqmc_seed(dqmc_handle *h, int32_t walker, uint64_t seed);
/* number of uniforms consumed so far by a walker */
int dqmc_uniforms_used(dqmc_handle *h, int32_t walker, uint64_t *used);
/* mc.s.current_slice, mc.s.direction (stack.jl:30-31); identical for all walkers */
int dqmc_get_state(dqmc_handle *h, int32_t *current_slice, int32_t *direction);

/* ---- the sweep loop ---------------------------------------------------- */
/* init!, build_stack, propagate (DQMC.jl:412-414) */
int dqmc_prepare(dqmc_handle *h);
/* build_stack only (stack.jl:242-255) */
int dqmc_build_stack(dqmc_handle *h);
/* propagate(mc) (stack.jl:502-631) */
int dqmc_propagate(dqmc_handle *h);
/* sweep_spatial(mc) (DQMC.jl:546-582) with propose_local / accept_local!
 * (HubbardModelAttractive.jl:113-155, HubbardModelRepulsive.jl:128-232) */
int dqmc_sweep_spatial(dqmc_handle *h);
/* update(mc, i) = propagate + sweep_spatial (DQMC.jl:523-538) */
int dqmc_update(dqmc_handle *h);
/* n_sweeps x (2*slices x update), the loop body of run! (DQMC.jl:420-437)
 * without measurements.  Asynchronous work is complete on return. */
int dqmc_sweep(dqmc_handle *h, int32_t n_sweeps);
/* updates until current_slice == 1 && direction == +1 — the measurement point
 * of run! (DQMC.jl:425-436); *n_updates receives the number of updates run */
int dqmc_update_until_measure(dqmc_handle *h, int32_t *n_updates);
/* wait for all queued device work of this handle */
int dqmc_synchronize(dqmc_handle *h);

/* ---- Green's functions -------------------------------------------------- */
/* mc.s.greens: effective G at current_slice, n_blocks x (n x n) */
int dqmc_get_greens_eff(dqmc_handle *h, int32_t walker, double *out);
int dqmc_set_greens_eff(dqmc_handle *h, int32_t walker, const double *in);
/* greens(mc) = eTinv * mc.s.greens * eT (DQMC.jl:721-730) */
int dqmc_get_greens(dqmc_handle *h, int32_t walker, double *out);
/* calculate_greens(mc, slice) from scratch (stack.jl:422-480); runs for all
 * walkers, returns the chosen walker's G.  Overwrites Ul..Tr, curr_U, tmp1/2
 * like the reference; does not touch mc.s.greens or the stack slots. */
int dqmc_calculate_greens_at(dqmc_handle *h, int32_t walker, int32_t slice, double *out);
/* the per-configuration step of replay!(mc) (DQMC.jl:647-653): calculate_greens(mc, slice) from
 * scratch into mc.s.greens for every walker (replay! uses slice = 0), current_slice <- 1 */
int dqmc_replay_greens(dqmc_handle *h, int32_t slice);
/* wrap_greens!(mc, mc.s.greens, slice, direction) on all walkers (stack.jl:491-500) */
int dqmc_wrap_greens(dqmc_handle *h, int32_t slice, int32_t direction);

/* ---- analysis ----------------------------------------------------------- */
int dqmc_get_stats(dqmc_handle *h, int32_t walker, dqmc_stats *out);

/* ---- measurement accumulators (stand-in for push!(LogBinner, greens(mc)),
 * measurements/generic.jl:207-215,260-263).  dqmc_accumulate_greens adds, for
 * every walker of this handle, the true G, G.^2 and the occupation 1-G_ii into
 * device-side sums.  Layout of the accumulator (doubles):
 *   [0 .. B*n*n)            sum G          (B = n_blocks)
 *   [B*n*n .. 2*B*n*n)      sum G.^2
 *   [2*B*n*n .. +B*n)       sum (1 - G_ii)
 *   last                    number of samples                                 */
int dqmc_accumulate_greens(dqmc_handle *h);
int dqmc_accumulator_size(dqmc_handle *h, size_t *n_doubles);
int dqmc_reset_accumulators(dqmc_handle *h);
/* copy accumulators to a host buffer, or device-to-device into a caller-owned
 * device buffer (e.g. a torch tensor that is then all-reduced over RCCL) */
int dqmc_get_accumulators(dqmc_handle *h, double *host_out);
int dqmc_export_accumulators(dqmc_handle *h, void *device_out);

/* ---- equal-time correlation measurements on the device ----------------------
 * cdc_kernel, sdc_{x,y,z}_kernel over EachSitePairByDistance and m{x,y,z}_kernel over EachSite
 * (measurements/measurements.jl:51-190, generic.jl:325-330; HubbardModelAttractive.jl:219-246).
 * dir_of[src + n*trg] (0-based) is the direction index of a pair as produced by
 * EachSitePairByDistance(lattice) (src/lattices/lattice_iterators.jl:157-190).  Accumulator layout:
 *   [cdc n_dirs][sdc_x n_dirs][sdc_y n_dirs][sdc_z n_dirs][mx n][my n][mz n][samples]
 * every pair quantity already divided by n_sites as finish! does (generic.jl:283-286);
 * dqmc_reset_accumulators clears these sums too. */
int dqmc_set_pair_directions(dqmc_handle *h, const int32_t *dir_of, int32_t n_dirs);
int dqmc_accumulate_correlations(dqmc_handle *h);
int dqmc_correlations_size(dqmc_handle *h, size_t *n_doubles);
int dqmc_get_correlations(dqmc_handle *h, double *host_out);
int dqmc_export_correlations(dqmc_handle *h, void *device_out);

/* pc_kernel over EachLocalQuadByDistance{K} (measurements/measurements.jl:199-214, generic.jl:287-290,
 * 341-349, src/lattices/lattice_iterators.jl:258-318; HubbardModelAttractive.jl:243-245):
 * trg_of[src + n*k] (0-based, -1 = none) is the site reached from src in the k-th shortest direction,
 * k < K, i.e. the (dir, trg) lists the iterator builds from EachSitePairByDistance.  Accumulator layout:
 * [n_dirs x K x K] in Julia's column-major order of output[dir12, dir1, dir2], already divided by
 * n_sites, then [samples].  Requires dqmc_set_pair_directions. */
int dqmc_set_local_targets(dqmc_handle *h, const int32_t *trg_of, int32_t K);
int dqmc_accumulate_pairing(dqmc_handle *h);
int dqmc_pairing_size(dqmc_handle *h, size_t *n_doubles);
int dqmc_get_pairing(dqmc_handle *h, double *host_out);
int dqmc_export_pairing(dqmc_handle *h, void *device_out);

/* ---- unequal-time Green's functions -------------------------------------------
 * UnequalTimeStack and its users (src/flavors/DQMC/unequal_time_stack.jl), for all walkers of the
 * handle at once; the sweep state (mc.s.greens, the DQMC stack, current_slice) is not disturbed.
 * The stacks are rebuilt lazily after the HS field has changed (the role of mc.last_sweep, :164-169).
 * Results stay on the device in three n x n x units buffers: `which` = 0 holds G(k,l) of
 * dqmc_ut_greens / the GreensIterator, or G0l of the CombinedGreensIterator; 1 = Gl0; 2 = Gll. */
/* build_stack(mc, mc.ut_stack) (:106-160) */
int dqmc_ut_build_stack(dqmc_handle *h);
/* stack inspection for tests (test/flavortests_DQMC.jl:75-96): which 0 forward, 1 backward, 2 inverse;
 * idx 0-based slot; per walker nb*n*n (U, T) and nb*n (D) doubles */
int dqmc_ut_get_stack(dqmc_handle *h, int32_t w, int32_t which, int32_t idx, double *U, double *D, double *T);
/* calculate_greens(mc, slice1, slice2) (:288-303; full1 :447-530, full2 :534-605) with 0 <= slices <=
 * slices; effective != 0 returns the stack's effective G, 0 applies _greens! (DQMC.jl:721-730) like
 * greens(mc, k, l) (:260-287) */
int dqmc_ut_greens(dqmc_handle *h, int32_t slice1, int32_t slice2, int32_t effective);
int dqmc_ut_get(dqmc_handle *h, int32_t w, int32_t which, double *host_out);
int dqmc_ut_export(dqmc_handle *h, int32_t which, void *device_out /* units*n*n doubles */);
/* GreensIterator(mc, :, l, recalculate) (:644-715): begin computes G(l <- l); each next advances k by one
 * and returns it in *k (-1 when exhausted); result in buffer 0 */
int dqmc_greens_iterator_begin(dqmc_handle *h, int32_t l, int32_t recalculate);
int dqmc_greens_iterator_next(dqmc_handle *h, int32_t *k);
/* CombinedGreensIterator(mc, recalculate) (:746-883): needs current_slice == 1; each next returns
 * l = 1..slices in *l (-1 when exhausted) with (G0l, Gl0, Gll) in buffers 0, 1, 2 */
int dqmc_combined_iterator_begin(dqmc_handle *h, int32_t recalculate);
int dqmc_combined_iterator_next(dqmc_handle *h, int32_t *l);

/* Susceptibilities on the device: apply!(::CombinedGreensIterator, ...) (measurements/generic.jl:226-243)
 * for charge_density_susceptibility, spin_density_susceptibility(:x, :y, :z) and, when
 * dqmc_set_local_targets has been called, pairing_susceptibility (measurements.jl:57-58,142-144,207;
 * packed kernels :76-92,158-192,215-219; HubbardModelAttractive.jl:226-249): the sum over l = 1..slices
 * of kernel(G00, G0l, Gl0, Gll), times delta_tau / n_sites as finish! does.  Needs current_slice == 1
 * and dqmc_set_pair_directions.  Accumulator layout:
 *   [cds n_dirs][sds_x n_dirs][sds_y n_dirs][sds_z n_dirs][ps n_dirs x K x K][samples]
 * dqmc_reset_accumulators clears these sums too. */
int dqmc_accumulate_susceptibilities(dqmc_handle *h, int32_t recalculate);
int dqmc_susceptibilities_size(dqmc_handle *h, size_t *n_doubles);
int dqmc_get_susceptibilities(dqmc_handle *h, double *host_out);
int dqmc_export_susceptibilities(dqmc_handle *h, void *device_out);

/* ---- measurement reduction over ranks -------------------
 * One process (or thread) per GPU; walkers never interact, the only collective is the reduction of the measurement
 * sums every `measure_rate` sweeps (DQMC.jl:429-436).  dqmc_reduce packs EVERY accumulator of the handle (Green's
 * function sums, correlations, pairing, susceptibilities - whichever are configured) and the DQMCAnalysis counters
 * of its walkers (prop_local, acc_local, MagnitudeStats sum / count / max / min, DQMC.jl:4-47) into one device
 * buffer [sums | 2 maxima | 2 minima] and runs ncclAllReduce (sum, max, min) over RCCL on the handle's stream.
 * The global sums are read with dqmc_get_reduced (section by section, same layouts and sizes as the local getters)
 * and the global counters with dqmc_get_reduced_stats.  The handle's own accumulators are NOT modified - they keep
 * the local sums - so the reduction may be repeated at every measurement interval (each call reduces the sums
 * accumulated so far since the last dqmc_reset_accumulators).
 * comm == NULL reduces over the walkers of this handle only.  A host that brings its own collective (MPI from
 * Julia, gloo in this repository's tests) uses dqmc_reduce_export -> reduce (sums, then maxima, then minima; layout
 * above) -> dqmc_reduce_import instead. */
typedef struct dqmc_comm dqmc_comm;
/* ncclGetUniqueId on rank 0 (128 bytes), to be broadcast by the host's own means */
int dqmc_comm_unique_id(void *id128);
/* ncclCommInitRank on device_id (collective: every rank calls it with the same id) */
int dqmc_comm_init(const void *id128, int32_t nranks, int32_t rank, int32_t device_id, dqmc_comm **out);
int dqmc_comm_destroy(dqmc_comm *c);
int dqmc_reduce(dqmc_handle *h, dqmc_comm *comm);
int dqmc_reduce_size(dqmc_handle *h, size_t *n_doubles /* sums + 4 */);
int dqmc_reduce_export(dqmc_handle *h, double *host_out);
int dqmc_reduce_import(dqmc_handle *h, const double *host_in);
enum { DQMC_RED_GREENS = 0, DQMC_RED_CORRELATIONS = 1, DQMC_RED_PAIRING = 2, DQMC_RED_SUSCEPTIBILITIES = 3 };
int dqmc_get_reduced(dqmc_handle *h, int32_t which, double *host_out);
int dqmc_get_reduced_stats(dqmc_handle *h, dqmc_stats *out);

/* ---- batched linalg primitives (unit parity with test/slice_matrices.jl) --
 * host in / host out, `batch` independent n x n problems, run on device_id.  */
/* vmul! family (src/linalg/general.jl:7-56): C = op(A)*op(B); transa/transb 0|1 */
int dqmc_vmul(int32_t device_id, int32_t n, int32_t batch, int32_t transa, int32_t transb,
              const double *A, const double *B, double *C);
/* udt_AVX_pivot!(U, D, T, pivot, temp, Val(apply)) (src/linalg/UDT.jl:192-306);
 * T holds the input on entry and T on exit.  The reference's contracts hold (test/slice_matrices.jl:202-234:
 * U unitary, U*Diagonal(D)*T = input resp. U*D*UpperTriangular(T)*P = input with P[i, pivot[i]] = 1).  At n = 256 and up to
 * 32 matrices the pivot order is the descending order of the INPUT's column norms (one-launch blocked factorisation), not the
 * step-by-step search of UDT.jl:212-246: D is then not sorted; DQMC_QR_NOBLOCKED=1 selects the reference's rule. */
int dqmc_udt_pivot(int32_t device_id, int32_t n, int32_t batch, double *U, double *D, double *T,
                   int64_t *pivot, int32_t apply_pivot);
/* rdivp!(A, T, O, pivot) (src/linalg/general.jl:138-166) */
int dqmc_rdivp(int32_t device_id, int32_t n, int32_t batch, double *A, const double *T,
               const int64_t *pivot);
/* calculate_greens_AVX! (src/flavors/DQMC/stack.jl:337-393) */
int dqmc_calculate_greens(int32_t device_id, int32_t n, int32_t batch, const double *Ul,
                          const double *Dl, const double *Tl, const double *Ur, const double *Dr,
                          const double *Tr, double *G);

/* CheckerboardTrue (DQMC(m; checkerboard=true), DQMC.jl:250-263) with the bond-group factors kept SPARSE on the
 * device: chkr_hop_half[g], chkr_hop[1], their inverses (and the adjoints, as separate factors) of
 * init_checkerboard_matrices (stack.jl:185-235) in ELL form - vals / cols [n_mats][n_sites][kmax], 0-based columns,
 * padding entries val = 0 - plus chkr_mu / chkr_mu_inv per block [n_blocks][n_sites] and seven factor sequences
 * (each up to 32 indices into the factor list, applied first to last):
 *   0 B X   (multiply_slice_matrix_left!, slice_matrices.jl:104-124)     3 X B    (:125-149)
 *   1 B^-1 X (:150-171)                                                   4 X B^-1 (:172-196)
 *   2 B' X  (multiply_daggered_slice_matrix_left!, :197-222)              5 X eT, 6 eTinv X (greens(), DQMC.jl:731-750)
 * Right products take the factors' transposes (rows of H' = columns of H).  After this call the propagation path
 * applies these sequences slab by slab in LDS instead of dense GEMMs with the multiplied-out constants (which
 * dqmc_create still needs: the unequal-time path uses them). */
int dqmc_set_checkerboard(dqmc_handle *h, int32_t kmax, int32_t n_mats, const double *vals, const int32_t *cols,
                          const double *mu, const double *mu_inv, const int32_t *seqs, const int32_t *lens);

/* The cooperative QR with the reference's pivot rule (n < 256, or 33..64 units at n = 256; 8 workgroups per matrix, bounded
 * hand-off spins) leaves its input intact; if a launch times out (CUs held by another stream for longer than the spins allow),
 * the guarded single-workgroup kernel launched behind it redoes the factorisation, so results stay valid.  This counter reports
 * how often that happened.  (The one-launch UDT has no second path: its time-outs fail the call, see dqmc_device_errors.) */
int dqmc_qr_fallbacks(dqmc_handle *h, int64_t *count);
/* which call sites of udt_AVX_pivot! (UDT.jl:192-306) this handle serves with the one-launch pre-pivoted factorisation:
 * bit 0 add_slice_sequence_left/right (stack.jl:272-311) and other callers, bit 1 / bit 2 the two factorisations of
 * calculate_greens_AVX! (stack.jl:349, :376); 0 = the reference's pivot rule everywhere (n != 256, > 32 units, DQMC_QR_NOBLOCKED) */
int dqmc_udt_one_launch_sites(dqmc_handle *h, int32_t *mask);
/* device error word as it stands (0 = no bounded wait inside a kernel has run out); no reference counterpart: the reference
 * has no concurrent workgroups (diagnostic next to DQMCAnalysis, src/flavors/DQMC/DQMC.jl:35-47) */
int dqmc_device_errors(dqmc_handle *h, int32_t *word);
/* commit the library was built from ("<short hash>[+]"); used by bench.py to tell fresh profile files from stale ones */
const char *dqmc_build_commit(void);
/* hash of the kernel sources the library was built from (csrc/Makefile: SOURCE_HASH); commits that leave the kernels alone
 * leave it alone - this is what decides whether a committed profile still describes the library */
const char *dqmc_build_source_hash(void);

/* ---- instrumentation ------------------------------------------------------ */
/* Per-kernel-family device time accumulated with HIP events on the handle's
 * stream when enabled (off by default; used by bench.py's roofline leg). */
enum { DQMC_K_GEMM = 0, DQMC_K_QR = 1, DQMC_K_TRSM = 2, DQMC_K_SWEEP = 3, DQMC_K_MISC = 4, DQMC_K_FLUSH = 5,
       DQMC_K_COUNT = 6 };
int dqmc_timing_enable(dqmc_handle *h, int32_t on);
int dqmc_timing_get(dqmc_handle *h, double *ms /* DQMC_K_COUNT */, int64_t *launches /* DQMC_K_COUNT */);
/* fp64 MFMA micro-benchmark: issues `iters` dependent-free v_mfma_f64_16x16x4_f64
 * per wave on every CU, returns achieved TFLOP/s (confirms the roofline peak) */
int dqmc_mfma_f64_peak(int32_t device_id, int32_t iters, double *tflops);

#ifdef __cplusplus
}
#endif
#endif
