/*
 * dqmc_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the DQMC hot path of
 * ffreyer/MonteCarlo.jl (src/flavors/DQMC + src/linalg + Hubbard models +
 * SquareLattice).  It exists to CHECK the HIP product path; nothing in the
 * product (montecarlo.jl_amd/, libdqmc_hip.so) may call, link or import it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: the reference is Julia and cannot be executed in this
 * pipeline (no julia binary, no network).  The oracle is pinned instead by the
 * reference's own fixtures: the bit-exact SquareLattice(4) checkerboard table
 * (test/flavortests_DQMC.jl:22-24), the algebraic contracts of
 * test/slice_matrices.jl:141-235, the independent LAPACK-dgeqp3 Green's
 * function oracle of test/testfunctions.jl:80-118 (restated with scipy), the
 * statistical goldens of test/integration_tests.jl:29-94 and the exact
 * diagonalisation of test/ED (restated in numpy).  See tests/test_oracle_*.py.
 *
 * Conventions: column-major, 0-based inside; integer tables that are compared
 * with reference fixtures are exported 1-based.  All citations are relative to
 * /root/reference.
 */
#ifndef DQMC_ORACLE_H
#define DQMC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lattices: src/lattices/square.jl:25-60, src/lattices/abstract.jl:99-115 */
void orc_square_neighs(int L, int64_t *neighs /* 4 x L^2, 1-based, col-major */);
void orc_square_bonds(int L, int64_t *bonds /* 2L^2 x 3 col-major: src,trg,type */);
/* src/flavors/DQMC/abstract.jl:23-54 ; returns n_groups. checkerboard is 3 x n_bonds
 * (col-major, 1-based), group_start/group_end are 1-based inclusive ranges. */
int orc_build_checkerboard(int n_sites, int n_bonds, const int64_t *bonds,
                           int64_t *checkerboard, int64_t *group_start,
                           int64_t *group_end, int max_groups);

/* ---- hopping matrices: HubbardModelAttractive.jl:78-91, HubbardModelRepulsive.jl:87-100 */
void orc_hopping_square(int L, double t, double mu, double *T /* n x n */);

/* ---- linalg: src/linalg/general.jl:7-166 */
/* timing-only: route the four dense products to a Fortran-interface dgemm (NULL restores the literal loops) */
void orc_set_dgemm_hook(void *fn);
void orc_vmul_nn(int n, double *C, const double *A, const double *B);
void orc_vmul_nt(int n, double *C, const double *A, const double *B); /* A * B'  */
void orc_vmul_tn(int n, double *C, const double *A, const double *B); /* A' * B  */
void orc_vmul_tt(int n, double *C, const double *A, const double *B); /* A' * B' */
void orc_vmul_nd(int n, double *C, const double *A, const double *d); /* A * Diagonal(d) */
void orc_vmul_dn(int n, double *C, const double *d, const double *B); /* Diagonal(d) * B */
void orc_rdivp(int n, double *A, const double *T, double *O, const int64_t *pivot /* 1-based */);
/* src/linalg/UDT.jl:192-306 ; pivot is 1-based on output */
/* study switch (see dqmc_oracle.c): pre-sorted pivot order per call site; 0 = reference rule */
void orc_set_udt_presort(int mask);
void orc_udt_pivot(int n, double *U, double *D, double *T, int64_t *pivot,
                   double *temp, int apply_pivot);
/* src/flavors/DQMC/stack.jl:337-393 ; overwrites all inputs */
void orc_calculate_greens(int n, double *Ul, double *Dl, double *Tl, double *Ur,
                          double *Dr, double *Tr, double *G, int64_t *pivot,
                          double *temp);

/* ---- RNG (not in the reference: Julia's global MersenneTwister cannot be
 * reproduced here; both oracle and product consume either a caller-supplied
 * uniform stream or this Philox4x32-10 counter stream with the reference's
 * conditional-consumption rule, src/flavors/DQMC/DQMC.jl:573) */
double orc_philox_uniform(uint64_t seed, uint64_t index);

/* ---- the DQMC object: DQMC.jl:133-189, stack.jl:1-85 */
typedef struct orc_mc orc_mc;

typedef struct {
    double max, min, sum; /* log10 magnitudes, DQMC.jl:4-31 */
    int64_t count;
} orc_magstats;

typedef struct {
    int64_t prop_local, acc_local; /* DQMC.jl:36-47 */
    orc_magstats imaginary_probability, negative_probability, propagation_error;
} orc_stats;

enum { ORC_ATTRACTIVE = 0, ORC_REPULSIVE = 1 };

/* eT.. are n_blocks consecutive n x n matrices (stack.jl:167-181 computed by caller) */
orc_mc *orc_create(int n_sites, int model_kind, int slices, int safe_mult,
                   double delta_tau, double U, const double *eT,
                   const double *eTinv, const double *eT2, const double *eTinv2,
                   int check_propagation_error, int check_sign_problem);
void orc_destroy(orc_mc *mc);
int orc_nblocks(const orc_mc *mc);
void orc_set_conf(orc_mc *mc, const int8_t *conf /* n_sites x slices col-major */);
void orc_get_conf(const orc_mc *mc, int8_t *conf);
void orc_set_uniforms(orc_mc *mc, const double *u, size_t n); /* copies */
size_t orc_uniforms_used(const orc_mc *mc);
void orc_seed(orc_mc *mc, uint64_t seed); /* philox mode, index reset to 0 */

void orc_init_stack(orc_mc *mc);            /* stack.jl:108-159 */
void orc_build_stack(orc_mc *mc);           /* stack.jl:242-255 */
void orc_propagate(orc_mc *mc);             /* stack.jl:502-631 */
void orc_sweep_spatial(orc_mc *mc);         /* DQMC.jl:546-582 */
void orc_update(orc_mc *mc);                /* DQMC.jl:523-538 */
void orc_prepare(orc_mc *mc);               /* DQMC.jl:412-414: init!, build_stack, propagate */
void orc_sweeps(orc_mc *mc, int n_sweeps);  /* DQMC.jl:420-437 without measurements */
/* run updates until current_slice==1 && direction==+1 (the measurement point,
 * DQMC.jl:425-436); returns number of updates done */
int orc_update_until_measure(orc_mc *mc);
void orc_wrap_greens(orc_mc *mc, double *gf /* n_blocks*n*n */, int slice /*1-based*/, int direction);
void orc_calculate_greens_at(orc_mc *mc, int slice, double *out); /* stack.jl:422-480 */
void orc_get_greens_eff(const orc_mc *mc, double *out);  /* mc.s.greens, n_blocks*n*n */
void orc_set_greens_eff(orc_mc *mc, const double *in);
void orc_get_greens(orc_mc *mc, double *out);            /* DQMC.jl:721-730 */
void orc_slice_matrix(orc_mc *mc, int slice /*1-based*/, double power, double *out); /* slice_matrices.jl:10-21 */
int orc_current_slice(const orc_mc *mc);
int orc_direction(const orc_mc *mc);
void orc_get_stats(const orc_mc *mc, orc_stats *st);

/* ---- config 1 plumbing: classical 2D Ising Metropolis,
 * src/flavors/MC/MC.jl:316-333, src/models/Ising/IsingModel.jl:83-101,177-185 */
typedef struct {
    double E, E2, M, M2; /* per-sweep sums over the measured sweeps */
    int64_t n_meas, accepted, proposed;
} orc_ising_result;
void orc_ising_run(int L, double beta, int thermalization, int sweeps,
                   uint64_t seed, int8_t *conf_inout /* L*L, may be NULL */,
                   orc_ising_result *res);

#ifdef __cplusplus
}
#endif
#endif
