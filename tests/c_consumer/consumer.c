/*
 * consumer.c — a plain C99 user of include/dqmc_hip.h ("a C test and Python ctypes").
 *
 * Shows that the header is valid C (not only C++), that its structs have the layout the library expects without anybody
 * re-declaring them by hand, and that the sweep path can be driven from compiled code the way a Julia `ccall` would:
 * create -> set_conf -> set_uniforms -> prepare -> sweep -> read back.  Test infrastructure; built and run by
 * tests/test_c_consumer.py, which compares what it writes with the CPU oracle.
 *
 *   consumer --probe                 no device work: device count, build commit, the error path of dqmc_create
 *   consumer <problem.bin> <out.bin> run the problem (format below, all little endian, written by the test)
 *
 * problem.bin: int32 n_sites, model_kind, slices, safe_mult, n_walkers, n_sweeps, n_uniforms, n_blocks;
 *              double delta_tau, U; eT, eTinv, eT2, eTinv2 (n_blocks*n*n doubles each);
 *              per walker: conf (n*slices int8), uniforms (n_uniforms doubles)
 * out.bin:     int32 current_slice, direction; per walker: conf, G_eff (n_blocks*n*n doubles), uint64 uniforms used,
 *              int64 prop_local, acc_local
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dqmc_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != DQMC_OK) {                                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dqmc_last_error(h));                   \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)

static int read_all(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

static int probe(void)
{
    dqmc_handle *h = NULL;
    dqmc_params p;
    const char *commit = dqmc_build_commit();
    int rc;
    memset(&p, 0, sizeof p);
    printf("devices %d\n", dqmc_device_count());
    printf("commit %s\n", commit ? commit : "(null)");
    printf("sizeof dqmc_params %u dqmc_stats %u\n", (unsigned)sizeof(dqmc_params), (unsigned)sizeof(dqmc_stats));
    /* slices not divisible by safe_mult (stack.jl:115) and no matrices: must be refused before any device call */
    p.n_sites = 4; p.slices = 7; p.safe_mult = 2; p.n_walkers = 1; p.delta_tau = 0.1; p.U = 1.0;
    rc = dqmc_create(&p, &h);
    printf("create(invalid) %d handle %s message \"%s\"\n", rc, h ? "set" : "null", dqmc_last_error(NULL));
    return (rc == DQMC_ERR_INVALID && h == NULL && commit && commit[0]) ? 0 : 1;
}

int main(int argc, char **argv)
{
    int32_t hd[8];
    double sc[2];
    dqmc_params p;
    dqmc_handle *h = NULL;
    dqmc_stats st;
    FILE *f, *o;
    double *mats, *u, *G;
    int8_t *conf;
    size_t nn, nconf;
    int32_t w, cs = 0, dir = 0;
    uint64_t used = 0;

    if (argc == 2 && strcmp(argv[1], "--probe") == 0) return probe();
    if (argc != 3) {
        fprintf(stderr, "usage: %s --probe | <problem.bin> <out.bin>\n", argv[0]);
        return 64;
    }
    f = fopen(argv[1], "rb");
    if (!f || read_all(f, hd, sizeof hd) || read_all(f, sc, sizeof sc)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    nn = (size_t)hd[7] * (size_t)hd[0] * (size_t)hd[0];
    nconf = (size_t)hd[0] * (size_t)hd[2];
    mats = (double *)malloc(4 * nn * sizeof(double));
    conf = (int8_t *)malloc(nconf);
    u = (double *)malloc(((size_t)hd[6] + 1) * sizeof(double));
    G = (double *)malloc(nn * sizeof(double));
    if (!mats || !conf || !u || !G || read_all(f, mats, 4 * nn * sizeof(double))) return 1;

    memset(&p, 0, sizeof p);
    p.n_sites = hd[0]; p.model_kind = hd[1]; p.slices = hd[2]; p.safe_mult = hd[3]; p.n_walkers = hd[4];
    p.device_id = 0; p.check_propagation_error = 1; p.check_sign_problem = 1;
    p.delta_tau = sc[0]; p.U = sc[1];
    p.eT = mats; p.eTinv = mats + nn; p.eT2 = mats + 2 * nn; p.eTinv2 = mats + 3 * nn;
    CHECK(dqmc_create(&p, &h));
    for (w = 0; w < hd[4]; ++w) {
        if (read_all(f, conf, nconf) || read_all(f, u, (size_t)hd[6] * sizeof(double))) return 1;
        CHECK(dqmc_set_conf(h, w, conf));
        CHECK(dqmc_set_uniforms(h, w, u, (size_t)hd[6]));
    }
    fclose(f);
    CHECK(dqmc_prepare(h));
    CHECK(dqmc_sweep(h, hd[5]));
    CHECK(dqmc_get_state(h, &cs, &dir));

    o = fopen(argv[2], "wb");
    if (!o) return 1;
    fwrite(&cs, sizeof cs, 1, o);
    fwrite(&dir, sizeof dir, 1, o);
    for (w = 0; w < hd[4]; ++w) {
        CHECK(dqmc_get_conf(h, w, conf));
        CHECK(dqmc_get_greens_eff(h, w, G));
        CHECK(dqmc_uniforms_used(h, w, &used));
        CHECK(dqmc_get_stats(h, w, &st));
        fwrite(conf, 1, nconf, o);
        fwrite(G, sizeof(double), nn, o);
        fwrite(&used, sizeof used, 1, o);
        fwrite(&st.prop_local, sizeof st.prop_local, 1, o);
        fwrite(&st.acc_local, sizeof st.acc_local, 1, o);
    }
    fclose(o);
    CHECK(dqmc_destroy(h));
    free(mats); free(conf); free(u); free(G);
    return 0;
}
