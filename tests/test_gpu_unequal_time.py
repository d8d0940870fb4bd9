"""unequal-time Green's functions on the device against oracle/unequal_time_oracle.py
and the properties of the reference's own test (test/flavortests_DQMC.jl:75-162)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def UT():
    from oracle import unequal_time_oracle
    return unequal_time_oracle


def _pair(gpu, O, kind, L=4, beta=2.0, safe_mult=5, seed=21, walkers=2, sweeps=1):
    """device engine and one oracle chain per walker, advanced to the measurement point"""
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=beta, safe_mult=safe_mult, n_walkers=walkers, seed=seed)
    mc.prepare()
    for _ in range(sweeps):
        mc.update_until_measure()
    oracles = []
    for w in range(walkers):
        o = O.OracleDQMC(L, kind, beta=beta, safe_mult=safe_mult)
        rng = np.random.Generator(np.random.Philox(key=mc.seeds[w]))
        o.set_conf(gpu.rand_conf(rng, L * L, o.slices)); o.seed(mc.seeds[w])
        o.prepare()
        for _ in range(sweeps):
            o.update_until_measure()
        assert np.array_equal(o.conf(), mc.conf(w))
        oracles.append(o)
    return mc, oracles


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_stacks_and_greens_kl(gpu, O, UT, kind):
    mc, oracles = _pair(gpu, O, kind)
    M, s = mc.p.slices, mc.p.safe_mult
    nr = M // s
    mc.ut_build_stack()
    for w, o in enumerate(oracles):
        for b in range(mc.nb):
            ut = UT.UnequalTimeOracle(o, b)
            ut.build_stack()
            for idx in range(nr + 1):
                U, D, T = mc.ut_stack("forward", idx, w)
                # U D T is the contract (UDT.jl:192-306); factors themselves are unique up to signs
                assert np.abs((U[b] * D[b]) @ T[b] - (ut.fu[idx] * ut.fd[idx]) @ ut.ft[idx]).max() \
                    < 1e-10 * max(1.0, ut.fd[idx].max())
                assert np.allclose(D[b], ut.fd[idx], rtol=1e-10)
                U, D, T = mc.ut_stack("backward", idx, w)
                assert np.allclose(D[b], ut.bd[idx], rtol=1e-10)
            for idx in range(nr):
                U, D, T = mc.ut_stack("inverse", idx, w)
                assert np.allclose(D[b], ut.id[idx], rtol=1e-10)
    pairs = [(0, 0), (3, 0), (7, 2), (M, 0), (M, M), (12, 12), (11, 4), (0, 5), (2, 9), (6, M), (0, M), (13, 17), (5, 5), (10, 5)]
    for k, l in pairs:
        for w, o in enumerate(oracles):
            g_eff = mc.calculate_greens_kl(k, l, w)
            for b in range(mc.nb):
                ref = UT.UnequalTimeOracle(o, b).calculate_greens(k, l)
                assert np.abs(g_eff[b] - ref).max() < 1e-10 * max(1.0, np.abs(ref).max()), (k, l, w, b)
        g_all = mc.greens_kl(k, l)
        for w, o in enumerate(oracles):
            for b in range(mc.nb):
                ref = UT.UnequalTimeOracle(o, b).greens(k, l)
                assert np.abs(g_all[w][b] - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    mc.close()


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_reference_properties(gpu, O, kind):
    """flavortests_DQMC.jl:107-118 on the device alone"""
    mc, _ = _pair(gpu, O, kind, walkers=1)
    M = mc.p.slices
    for k in range(0, M + 1):
        g1 = mc.calculate_greens(k)
        g2 = mc.calculate_greens_kl(k, k)
        for b in range(mc.nb):
            assert np.abs(g1[b] - g2[b]).max() < 1e-12, k
    for k in range(0, M):
        a, c = mc.greens_kl(k, 0, 0), mc.greens_kl(k, M, 0)
        for b in range(mc.nb):
            assert np.abs(a[b] + c[b]).max() < 1e-12, k
    # the sweep state is untouched: the chain continues exactly like a chain that never measured
    mc2, _ = _pair(gpu, O, kind, walkers=1)
    mc.update_until_measure(); mc2.update_until_measure()
    assert np.array_equal(mc.conf(0), mc2.conf(0))
    mc.close(); mc2.close()


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_iterators(gpu, O, UT, kind):
    """flavortests_DQMC.jl:121-162: both iterators against greens(mc, k, l), high precision with
    recalculate = safe_mult, lower with 4 safe_mult; and against the oracle's iterators step by step"""
    mc, oracles = _pair(gpu, O, kind, walkers=2)
    M, s = mc.p.slices, mc.p.safe_mult
    w = 1
    Gk0 = [mc.greens_kl(k, 0, w) for k in range(M + 1)]
    G0k = [mc.greens_kl(0, k, w) for k in range(M + 1)]
    Gkk = [mc.greens_kl(k, k, w) for k in range(M + 1)]
    for recalc, tol in ((s, 1e-12), (4 * s, 2e-9)):
        out = list(mc.greens_iterator(0, recalc, walker=w))
        assert len(out) == M + 1
        for k, g in enumerate(out):
            for b in range(mc.nb):
                assert np.abs(g[b] - Gk0[k][b]).max() < tol, (recalc, k)
        out = list(mc.combined_greens_iterator(recalc, walker=w))
        assert len(out) == M
        for i, (g0l, gl0, gll) in enumerate(out):
            for b in range(mc.nb):
                assert np.abs(gl0[b] - Gk0[i + 1][b]).max() < tol, (recalc, i, "Gl0")
                assert np.abs(g0l[b] - G0k[i + 1][b]).max() < tol, (recalc, i, "G0l")
                assert np.abs(gll[b] - Gkk[i + 1][b]).max() < tol, (recalc, i, "Gll")
    # step-by-step parity with the oracle's restatement of the iterators (same recalculate)
    o = oracles[w]
    for b in range(mc.nb):
        ut = UT.UnequalTimeOracle(o, b)
        # recalculate = safe_mult: tight; 4 safe_mult (never recalculated at this beta): both sides drift
        for recalc, tol in ((s, 1e-11), (4 * s, 5e-9)):
            ref = list(ut.combined_greens_iterator(o.greens_eff()[b], recalc))
            dev = list(mc.combined_greens_iterator(recalc, walker=w))
            assert len(ref) == len(dev) == M
            for (r0, r1, r2), (d0, d1, d2) in zip(ref, dev):
                assert max(np.abs(d0[b] - r0).max(), np.abs(d1[b] - r1).max(), np.abs(d2[b] - r2).max()) < tol, recalc
        ref = list(ut.greens_iterator(3, s))
        dev = list(mc.greens_iterator(3, s, walker=w))
        assert len(ref) == len(dev) == M + 1 - 3
        for r, d in zip(ref, dev):
            assert np.abs(d[b] - r).max() < 1e-10
    mc.close()


def test_unequal_time_errors(gpu):
    mc = gpu.DQMC(gpu.HubbardModelAttractive(4, 2), beta=1.0, n_walkers=1)
    with pytest.raises(RuntimeError):
        mc.greens_kl(1, 0)          # not prepared
    mc.prepare()
    with pytest.raises(RuntimeError):
        mc.greens_kl(11, 0)         # slice out of range
    with pytest.raises(RuntimeError):
        list(mc.combined_greens_iterator())  # current_slice != 1 right after prepare
    mc.close()


def test_unequal_time_config2_size(gpu):
    """BASELINE config 2 shape (8x8, beta=4): symmetry G(t,0) = -G(t,beta) and G(k,k) at full depth"""
    mc = gpu.DQMC(gpu.HubbardModelAttractive(8, 2), beta=4.0, n_walkers=2, seed=3)
    mc.prepare(); mc.update_until_measure()
    M = mc.p.slices
    for k in (0, 7, 20, 33, M):
        g1, g2 = mc.calculate_greens(k, 1), mc.calculate_greens_kl(k, k, 1)
        assert np.abs(g1[0] - g2[0]).max() < 1e-11
    for k in (1, 15, 39):
        assert np.abs(mc.greens_kl(k, 0, 0)[0] + mc.greens_kl(k, M, 0)[0]).max() < 1e-11
    mc.close()


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_susceptibilities(gpu, O, R, UT, kind):
    """charge_density_/spin_density_/pairing_susceptibility summed on the device against the oracle's
    CombinedGreensIterator + the generic packed kernels (generic.jl:226-243)"""
    mc, oracles = _pair(gpu, O, kind, walkers=2)
    L, s = 4, mc.p.safe_mult
    it = gpu.EachLocalQuadByDistance(mc.model.l)
    mc.set_local_targets(it)
    mc.reset_accumulators()
    mc.accumulate_susceptibilities(recalculate=s)
    res = mc.susceptibilities()
    assert res["count"] == 2 and res["PS"].shape == (16, 5, 5)
    ref = None
    for o in oracles:
        uts = [UT.UnequalTimeOracle(o, b) for b in range(o.nb)]
        its = [u.combined_greens_iterator(o.greens_eff()[b], s) for b, u in enumerate(uts)]
        steps = [tuple([blk[q] for blk in per_block] for q in range(3)) for per_block in zip(*its)]
        r = R.susceptibilities(o.greens(), steps, L, kind == "attractive", 5, o.delta_tau)
        ref = r if ref is None else {k: ref[k] + r[k] for k in r}
    for k in ("CDS", "SDSx", "SDSy", "SDSz", "PS"):
        assert np.abs(res[k] - ref[k] / 2).max() < 1e-10 * max(1.0, np.abs(ref[k]).max()), k
    # a second sample adds up; reset clears
    mc.accumulate_susceptibilities(recalculate=s)
    assert mc.susceptibilities()["count"] == 4
    assert np.abs(mc.susceptibilities()["CDS"] - res["CDS"]).max() < 1e-12
    mc.reset_accumulators()
    mc.accumulate_susceptibilities()
    assert mc.susceptibilities()["count"] == 2
    mc.close()
