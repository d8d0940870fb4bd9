"""Pins oracle/unequal_time_oracle.py with the properties the reference's own test
asserts (test/flavortests_DQMC.jl:75-162) and with a dense evaluation of the definition."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def UT():
    from oracle import unequal_time_oracle
    return unequal_time_oracle


def _setup(O, kind, L=4, beta=2.0, safe_mult=5, seed=5):
    mc = O.OracleDQMC(L, kind, beta=beta, safe_mult=safe_mult)
    mc.set_conf(O.random_conf(seed, L * L, mc.slices)); mc.seed(seed)
    mc.prepare()
    mc.sweeps(2)
    # bring the chain to current_slice == 1, direction == +1 (measurement point, DQMC.jl:429)
    mc.update_until_measure()
    return mc


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_unequal_time_against_definition(O, UT, kind):
    mc = _setup(O, kind)
    M = mc.slices
    for b in range(mc.nb):
        ut = UT.UnequalTimeOracle(mc, b)
        for k, l in [(0, 0), (3, 0), (7, 2), (M, 0), (M, M), (12, 12), (11, 4), (0, 5), (2, 9), (6, M), (0, M), (13, 17)]:
            g = ut.calculate_greens(k, l)
            ref = UT.brute_force_greens(mc, b, k, l)
            # the dense reference inverts products of up to M slice matrices: it is the less accurate side
            assert np.abs(g - ref).max() < 5e-9, (kind, b, k, l, np.abs(g - ref).max())


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_unequal_time_reference_properties(O, UT, kind):
    """flavortests_DQMC.jl:107-118: G(k,k) from the unequal-time stack equals calculate_greens(mc, k);
    G(t, 0) = -G(t, beta)"""
    mc = _setup(O, kind)
    M = mc.slices
    for b in range(mc.nb):
        ut = UT.UnequalTimeOracle(mc, b)
        for k in range(0, M + 1):
            g1 = mc.calculate_greens_at(k)[b]
            g2 = ut.calculate_greens(k, k)
            assert np.abs(g1 - g2).max() < 1e-12, (k, np.abs(g1 - g2).max())
        for k in range(0, M):
            assert np.abs(ut.greens(k, 0) + ut.greens(k, M)).max() < 1e-12


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_unequal_time_iterators(O, UT, kind):
    """flavortests_DQMC.jl:121-162: GreensIterator and CombinedGreensIterator against greens(mc, k, l),
    high precision with recalculate = safe_mult, lower with 4 safe_mult"""
    mc = _setup(O, kind)
    M, s = mc.slices, mc.safe_mult
    assert mc.current_slice == 1
    for b in range(mc.nb):
        ut = UT.UnequalTimeOracle(mc, b)
        Gk0 = [ut.greens(k, 0) for k in range(M + 1)]
        G0k = [ut.greens(0, k) for k in range(M + 1)]
        Gkk = [mc.eTinv @ mc.calculate_greens_at(k)[b] @ mc.eT for k in range(M + 1)]
        for recalc, tol in ((s, 1e-13), (4 * s, 1e-11)):
            out = list(ut.greens_iterator(0, recalc))
            assert len(out) == M + 1
            for k, g in enumerate(out):
                assert np.abs(g - Gk0[k]).max() < tol, (recalc, k)
        g_eff = mc.greens_eff()[b]
        for recalc, tol in ((s, 1e-13), (4 * s, 2e-9)):  # 16 sites, no recalculation before l = M
            out = list(ut.combined_greens_iterator(g_eff, recalc))
            assert len(out) == M
            for i, (g0l, gl0, gll) in enumerate(out):
                l = i + 1
                assert np.abs(gl0 - Gk0[l]).max() < tol, (recalc, l, "Gl0")
                assert np.abs(g0l - G0k[l]).max() < tol, (recalc, l, "G0l")
                assert np.abs(gll - Gkk[l]).max() < tol, (recalc, l, "Gll")
        # GreensIterator with l > 0
        out = list(ut.greens_iterator(3, s))
        assert len(out) == M + 1 - 3
        for i, g in enumerate(out):
            assert np.abs(g - ut.greens(3 + i, 3)).max() < 1e-12


def test_unequal_time_reference_setup(O, UT):
    """the reference's own case (test/flavortests_DQMC.jl:75-162): HubbardModelAttractive(6, 1) = a
    6-site chain, beta = 15, safe_mult = 5 (150 slices), with the reference's tolerances"""
    L = 6
    T = np.zeros((L, L))
    for i in range(L):
        T[i, (i + 1) % L] = T[(i + 1) % L, i] = -1.0
    mc = O.OracleDQMC(L, "attractive", beta=15.0, safe_mult=5, hopping=np.asfortranarray(T))
    mc.set_conf(O.random_conf(11, L, mc.slices)); mc.seed(11)
    mc.prepare()
    mc.update_until_measure()
    assert mc.current_slice == 1
    M, s = mc.slices, mc.safe_mult
    ut = UT.UnequalTimeOracle(mc, 0)
    for k in range(0, M + 1):
        assert np.abs(mc.calculate_greens_at(k)[0] - ut.calculate_greens(k, k)).max() < 1e-13
    for k in range(0, M):
        assert np.abs(ut.greens(k, 0) + ut.greens(k, M)).max() < 1e-13
    Gk0 = [ut.greens(k, 0) for k in range(M + 1)]
    G0k = [ut.greens(0, k) for k in range(M + 1)]
    Gkk = [mc.eTinv @ mc.calculate_greens_at(k)[0] @ mc.eT for k in range(M + 1)]
    for recalc, tol in ((s, 1e-13), (4 * s, 1e-11)):
        for k, g in enumerate(ut.greens_iterator(0, recalc)):
            assert np.abs(g - Gk0[k]).max() < tol, (recalc, k)
    for recalc, tol in ((s, 1e-13), (4 * s, 1e-10)):
        for i, (g0l, gl0, gll) in enumerate(ut.combined_greens_iterator(mc.greens_eff()[0], recalc)):
            assert np.abs(gl0 - Gk0[i + 1]).max() < tol
            assert np.abs(g0l - G0k[i + 1]).max() < tol
            assert np.abs(gll - Gkk[i + 1]).max() < tol
