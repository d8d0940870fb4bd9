"""GPU parity of the DQMC sweep path against the CPU oracle on identical seeds:
HS field bit-exact, effective Green's function within 1e-10 relative."""
import os

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_pair(gpu, O, L, kind, beta, n_walkers=2, seed=123, **kw):
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2, **kw.pop("model_kw", {}))
    mc = gpu.DQMC(model, beta=beta, n_walkers=n_walkers, seed=seed, **kw)
    refs = []
    for w in range(n_walkers):
        mk = {}
        if kind == "attractive":
            mk["mu"] = model.mu
        o = O.OracleDQMC(L, kind, beta=beta, delta_tau=mc.p.delta_tau, safe_mult=mc.p.safe_mult, U=model.U, **mk)
        o.set_conf(mc.conf(w))
        o.seed(mc.seeds[w])
        refs.append(o)
    return mc, refs


def compare(mc, refs, tol=TOL, conf=True):
    for w, o in enumerate(refs):
        if conf:
            assert np.array_equal(mc.conf(w), o.conf()), "HS field of walker %d differs" % w
        for b, (g, g0) in enumerate(zip(mc.greens_eff(w), o.greens_eff())):
            e = relerr(g, g0)
            assert e < tol, "walker %d block %d: G rel err %g" % (w, b, e)


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_prepare_matches_oracle(gpu, O, kind):
    """build_stack + propagate (DQMC.jl:412-414): G at slice M"""
    mc, refs = make_pair(gpu, O, 4, kind, 1.0)
    mc.prepare()
    for o in refs:
        o.prepare()
    assert (mc.current_slice, mc.direction) == (refs[0].current_slice, refs[0].direction)
    compare(mc, refs)
    mc.close()


@pytest.mark.parametrize("kind,L,beta", [("attractive", 4, 1.0), ("repulsive", 4, 1.0), ("attractive", 8, 4.0)])
def test_stepwise_updates(gpu, O, kind, L, beta):
    """propagate / sweep_spatial one call at a time through more than a full sweep"""
    mc, refs = make_pair(gpu, O, L, kind, beta)
    mc.prepare()
    for o in refs:
        o.prepare()
    nupd = 2 * mc.p.slices + 5
    for u in range(nupd):
        mc.propagate()
        for o in refs:
            o.propagate()
        assert (mc.current_slice, mc.direction) == (refs[0].current_slice, refs[0].direction)
        compare(mc, refs, conf=False)
        mc.sweep_spatial()
        for o in refs:
            o.sweep_spatial()
        compare(mc, refs)
    for w, o in enumerate(refs):
        a, st = mc.analysis(w), o.stats()
        assert (a.prop_local, a.acc_local) == (st.prop_local, st.acc_local)
        assert mc.uniforms_used(w) == o.uniforms_used()
    mc.close()


def test_uniform_stream_mode(gpu, O):
    """host-supplied uniforms consumed with the reference's conditional rule (DQMC.jl:573)"""
    mc, refs = make_pair(gpu, O, 4, "attractive", 1.0, n_walkers=1)
    u = np.random.default_rng(5).random(4000)
    mc.set_uniforms(0, u)
    refs[0].set_uniforms(u)
    mc.prepare(); refs[0].prepare()
    mc.sweep(2); refs[0].sweeps(2)
    compare(mc, refs)
    assert mc.uniforms_used(0) == refs[0].uniforms_used()
    mc.close()


def test_cfg2_sweeps(gpu, O):
    """BASELINE config 2: attractive 8x8, beta=4 (n=64, M=40), 1 walker, 3 sweeps"""
    mc, refs = make_pair(gpu, O, 8, "attractive", 4.0, n_walkers=1)
    mc.prepare(); refs[0].prepare()
    mc.sweep(3); refs[0].sweeps(3)
    compare(mc, refs)
    mc.close()


@pytest.mark.parametrize("kind,safe_mult", [("attractive", 1), ("attractive", 2), ("attractive", 5), ("repulsive", 5)])
def test_n256_short_slice_chains(gpu, O, kind, safe_mult):
    """n = 256 with safe_mult below 10 (beta = 1: M = 10): the slab kernel runs chains of 1, 2 and 5 products (its request
    stream runs on into the next step's operand and wraps at the last step) and, for the repulsive model, two blocks per walker;
    a full sweep of updates, update by update, against the oracle"""
    mc, refs = make_pair(gpu, O, 16, kind, 1.0, n_walkers=2, safe_mult=safe_mult)
    mc.prepare()
    for o in refs:
        o.prepare()
    compare(mc, refs)
    for u in range(2 * mc.p.slices + 1):
        mc.update()
        for o in refs:
            o.update()
        assert (mc.current_slice, mc.direction) == (refs[0].current_slice, refs[0].direction)
        compare(mc, refs)
    mc.close()


def test_cfg3_one_sweep(gpu, O):
    """BASELINE config 3 shape: attractive 16x16, beta=8 (n=256, M=80), 2 walkers, 1 sweep"""
    mc, refs = make_pair(gpu, O, 16, "attractive", 8.0, n_walkers=2)
    mc.prepare()
    for o in refs:
        o.prepare()
    compare(mc, refs)
    mc.sweep(1)
    for o in refs:
        o.sweeps(1)
    compare(mc, refs)
    mc.close()


def test_true_greens_and_calculate_at(gpu, O):
    """greens(mc) = eTinv*G*eT (test/measurements.jl:162-184) and calculate_greens(mc, slice)
    (test/flavortests_DQMC.jl:62-69)"""
    mc, refs = make_pair(gpu, O, 4, "attractive", 1.0, n_walkers=1, safe_mult=5, model_kw=dict(mu=0.5))
    mc.prepare(); refs[0].prepare()
    assert relerr(mc.greens(0)[0], refs[0].greens()[0]) < TOL
    for k in (0, 1, 4, 5, 9, 10):
        assert relerr(mc.calculate_greens(k)[0], refs[0].calculate_greens_at(k)[0]) < TOL
    mc.close()


def test_cfg4_repulsive_shape(gpu, O):
    """BASELINE config 4 shape: repulsive 16x16, beta=8 (two 256x256 blocks sharing the HS field),
    one walker: prepare + a quarter sweep against the oracle"""
    mc, refs = make_pair(gpu, O, 16, "repulsive", 8.0, n_walkers=1)
    mc.prepare(); refs[0].prepare()
    compare(mc, refs)
    for _ in range(40):
        mc.update(); refs[0].update()
    compare(mc, refs)
    a, st = mc.analysis(0), refs[0].stats()
    assert (a.prop_local, a.acc_local) == (st.prop_local, st.acc_local)
    mc.close()


def test_cfg5_shape_n576(gpu, O):
    """BASELINE config 5 lattice (24x24, n=576, dtau=0.05) at a short beta: exercises the n > 256
    kernel paths (streaming QR, LDS-slab TRSM, 8-site sweep chunks)"""
    model = gpu.HubbardModelAttractive(24, 2)
    mc = gpu.DQMC(model, beta=1.0, delta_tau=0.05, n_walkers=1, seed=7)
    o = O.OracleDQMC(24, "attractive", beta=1.0, delta_tau=0.05)
    o.set_conf(mc.conf(0)); o.seed(mc.seeds[0])
    mc.prepare(); o.prepare()
    assert relerr(mc.greens_eff(0)[0], o.greens_eff()[0]) < TOL
    for _ in range(3):
        mc.update(); o.update()
    assert np.array_equal(mc.conf(0), o.conf())
    assert relerr(mc.greens_eff(0)[0], o.greens_eff()[0]) < TOL
    mc.close()


@pytest.mark.parametrize("env", [{}, {"DQMC_QR_NOPANEL": "1"}], ids=["panel_qr", "streaming_qr"])
def test_cfg5_full_depth_against_the_oracle(gpu, O, env):
    """BASELINE config 5 at its FULL depth against the oracle (stack.jl:242-255, 337-393): 24x24, beta = 20, dtau = 0.05
    (n = 576, 400 slices, a stack of K = 40 decompositions - the deep-stabilisation case the configuration exists for):
    prepare() (build_stack + the Green's function at slice M from a 40-deep stack) and the first safe_mult + 1 updates
    (one stabilisation step on the deep stack, on the way down).  This is where the panel QR's down-dated norms
    (qr_panel_kernel: the one departure from the reference's from-scratch norms, UDT.jl:151-168) would show if they
    picked worse pivots; the second parametrisation runs the streaming kernel with the reference's arithmetic.
    The oracle's dense products go through OpenBLAS here (same algorithm, other summation order; ~1-2 min of CPU)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    blas = O.use_openblas_dgemm(True)
    try:
        model = gpu.HubbardModelAttractive(24, 2)
        mc = gpu.DQMC(model, beta=20.0, delta_tau=0.05, n_walkers=1, seed=11)
        assert mc.p.slices == 400 and mc.p.safe_mult == 10
        o = O.OracleDQMC(24, "attractive", beta=20.0, delta_tau=0.05)
        o.set_conf(mc.conf(0)); o.seed(mc.seeds[0])
        mc.prepare(); o.prepare()
        e0 = relerr(mc.greens_eff(0)[0], o.greens_eff()[0])
        assert e0 < TOL, e0
        for _ in range(11):
            mc.update(); o.update()
        assert (mc.current_slice, mc.direction) == (o.current_slice, o.direction)
        assert np.array_equal(mc.conf(0), o.conf())
        e1 = relerr(mc.greens_eff(0)[0], o.greens_eff()[0])
        assert e1 < TOL, e1
        a, st = mc.analysis(0), o.stats()
        assert (a.prop_local, a.acc_local) == (st.prop_local, st.acc_local)
        assert a.propagation_error.count == st.propagation_error.count
        print("cfg5 depth: G error after prepare %.2e, after 11 updates %.2e (OpenBLAS oracle: %s)" % (e0, e1, blas))
        mc.close()
    finally:
        O.use_openblas_dgemm(False)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_ed_known_answer_on_gpu(gpu):
    """test/ED/ED_tests.jl:91-176 on the product path: mean Green's function of the 2x2 Hubbard
    models (U=1, t=1, beta=1, dtau=0.1, safe_mult=5; mu=1 for the attractive model) from run()
    against exact diagonalisation (fixture tests/golden/ed_hubbard_2x2.json), atol = rtol = 2*dtau^2"""
    import json, os
    ed = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ed_hubbard_2x2.json")))
    for model, key in ((gpu.HubbardModelRepulsive(2, 2, U=1.0), "repulsive_U1_t1"),
                       (gpu.HubbardModelAttractive(2, 2, U=1.0, mu=1.0), "attractive_U1_mu1_t1")):
        mc = gpu.DQMC(model, beta=1.0, safe_mult=5, n_walkers=64, seed=2024, thermalization=100, sweeps=150,
                      measure_rate=1)
        mc.run()
        res = mc.unpack_accumulators(mc.accumulators())
        assert res["count"] == 64 * 150
        Ged = np.array(ed[key])
        for b in range(mc.nb):
            ref = Ged[4 * b:4 * b + 4, 4 * b:4 * b + 4]
            assert np.all(np.abs(res["G"][b] - ref) <= 0.02 + 0.02 * np.abs(ref)), (key, b)
            assert np.all(np.abs(res["occupation"][b] - (1 - np.diag(ref))) <= 0.02 + 0.02 * np.abs(1 - np.diag(ref)))
        a = mc.analysis(0)
        assert a.prop_local == 250 * 20 * 4 and a.propagation_error.count == 0
        mc.close()


def test_repeated_udt_launches_are_bit_reproducible(gpu):
    """The cooperative QR tags its mailbox packets with a per-handle launch counter; drive one handle
    through several hundred launches (calculate_greens(mc, slice) = ~7 UDTs each, n = 16 and n = 64) and
    require bit-identical Green's functions every time: the pivot decisions may not depend on timing
    or on the launch counter (a 16-bit tag alias at launch 128 once did)."""
    for L, reps in ((4, 90), (8, 40)):
        mc = gpu.DQMC(gpu.HubbardModelRepulsive(L, 2), beta=1.0, safe_mult=5, n_walkers=3, seed=9)
        mc.prepare()
        first = [mc.calculate_greens(3, w) for w in range(3)]
        for _ in range(reps):
            for w in (0, 2):
                g = mc.calculate_greens(3, w)
                assert all(np.array_equal(g[b], first[w][b]) for b in range(2))
        mc.close()


@pytest.mark.parametrize("kind,L", [("attractive", 6), ("repulsive", 6), ("attractive", 10)])
def test_odd_sizes_padding_paths(gpu, O, kind, L):
    """n = 36 and n = 100 are multiples of neither 8, 16 nor 64: padded lanes in the sweep kernel, partial
    tiles in the GEMMs, partial blocks in QR / TRSM, the generic (non-flush-kernel) update flush"""
    mc, refs = make_pair(gpu, O, L, kind, 1.0, n_walkers=2, safe_mult=5)
    mc.prepare()
    for o in refs:
        o.prepare()
    compare(mc, refs)
    for _ in range(12):
        mc.update()
        for o in refs:
            o.update()
    compare(mc, refs)
    mc.close()


def test_cooperative_qr_timeout_falls_back(gpu, O):
    """A cooperative-QR launch whose hand-offs time out must not corrupt anything: the kernel leaves its input intact
    and the guarded single-workgroup kernel behind it redoes the factorisation.  DQMC_QR_FORCE_TIMEOUT makes every
    cooperative launch give up at once (read at the first launch, hence a child process)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import __graft_entry__ as g
        gpu = g.load_package(); O = g.load_oracle()
        mc = gpu.DQMC(gpu.HubbardModelAttractive(4, 2), beta=1.0, n_walkers=2, seed=11)
        ref = []
        for w in range(2):
            o = O.OracleDQMC(4, "attractive", beta=1.0); o.set_conf(mc.conf(w)); o.seed(mc.seeds[w]); o.prepare(); ref.append(o)
        mc.prepare()
        for _ in range(25):
            mc.update(); [o.update() for o in ref]
        for w, o in enumerate(ref):
            assert np.array_equal(mc.conf(w), o.conf())
            e = np.abs(mc.greens_eff(w)[0] - o.greens_eff()[0]).max() / np.abs(o.greens_eff()[0]).max()
            assert e < 1e-10, e
        n = mc.qr_fallbacks()
        assert n > 0, "the forced time-outs were not taken"
        print("fallbacks", n)
    """ % ROOT)
    env = dict(os.environ, DQMC_QR_FORCE_TIMEOUT="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "fallbacks" in p.stdout


@pytest.mark.parametrize("plain", [{"DQMC_NO_SLAB": "1"}, {"DQMC_SWEEP_SPLIT": "1"},
                                   {"DQMC_SWEEP_SPLIT": "1", "DQMC_FLUSH_NCP2": "1"}, {"DQMC_QR_SC1": "1"},
                                   {"DQMC_QR_NOBLOCKED": "1"}, {"DQMC_QR_NOBLOCKED": "1", "DQMC_QR_SC1": "1"},
                                   {"DQMC_QR_NOBLOCKED": "1", "DQMC_QR_TAIL": "0"},
                                   {"DQMC_QR_NOBLOCKED": "1", "DQMC_QR_NOCOOP": "1"},
                                   {"DQMC_QR_NOBLOCKED": "1", "DQMC_QR_NOCOOP": "1", "DQMC_QR_TAIL": "0"},
                                   {"DQMC_QRB_SITES": "1"},
                                   {"DQMC_TRSM_LL": "1"}, {"DQMC_TRSM_SIMPLE": "1"}, {"DQMC_SWEEP_OLD": "1"},
                                   {"DQMC_TRSM_BOUNDS": "1"},
                                   {"DQMC_QR_NOBLOCKED": "1", "DQMC_QR_NOCOOP": "1", "DQMC_QR_TILE_BOUNDS": "1"}],
                         ids=["slab_chains", "fused_sweep", "two_pass_flush", "blocked_udt_sc1_mailbox",
                              "udt_pivoted_cooperative", "udt_pivoted_cooperative_sc1", "udt_pivoted_cooperative_only",
                              "udt_pivoted_tile_plus_tail", "udt_pivoted_tile_only", "blocked_udt_slice_sequences_only",
                              "trsm_left_looking", "trsm_substitution",
                              "sweep_round1", "trsm_with_bounds", "udt_pivoted_tile_with_bounds"])
def test_fast_paths_against_their_plain_forms(gpu, plain):
    """The default launch forms at n = 256 (slab-resident product chains, elimination fused with the previous chunk's
    flush, the one-launch pre-pivoted UDT; the two-pass flush of the throughput regime against the one-pass one) against
    the forms they replace, selected per handle through the environment: same seeds, HS field identical, G within the
    parity tolerance.  The udt_pivoted_* entries run the reference's own pivot rule (UDT.jl:212-246: the cooperative
    two-phase QR, the tile kernels) in place of the pre-pivoted blocked factorisation - a different column order, the same
    G; *_sc1_* are the forms whose hand-off stores are agent-scope (write-through), i.e. inside the HIP memory model;
    the remaining entries are the fallbacks that ship behind switches (round-1 TRSM and sweep kernels)."""
    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            mc = gpu.DQMC(gpu.HubbardModelAttractive(16, 2), beta=2.0, n_walkers=2, seed=7)
            mc.prepare()
            mc.sweep(2)
            out = [mc.conf(w).copy() for w in range(2)], [mc.greens_eff(w)[0].copy() for w in range(2)]
            assert mc.qr_fallbacks() == 0  # (a silent fallback would make any cooperative form look correct)
            mc.close()
            return out
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    conf_fast, g_fast = run({})
    conf_plain, g_plain = run(plain)
    for w in range(2):
        assert np.array_equal(conf_fast[w], conf_plain[w])
        assert relerr(g_fast[w], g_plain[w]) < TOL
