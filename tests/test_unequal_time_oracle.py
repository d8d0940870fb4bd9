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


def test_time_displaced_conventions_against_ed(O, R, UT):
    """At U = 0 (lambda = 0: the slice matrices do not depend on the HS field) everything is exact and
    Wick's theorem holds, so exact diagonalisation pins
      * the sign / ordering conventions of G(l,0) = <c_i(tau) c_j^dag(0)> and G(0,l) = -<c_j^dag(tau) c_i(0)>
        (unequal_time_stack.jl:262-265, full2's rmul!(-1)), up to the Trotter-free hopping (exact here
        because [T, V] = 0 when V = 0),
      * the packed kernels cdc / sdc_z / sdc_x / pc (measurements.jl:76-92,158-192,215-219) against the
        imaginary-time correlators <A(tau) B(0)> = tr(rho e^{tau H} A e^{-tau H} B)."""
    L, N, beta, dtau = 2, 4, 1.0, 0.1
    mc = O.OracleDQMC(L, "repulsive", beta=beta, delta_tau=dtau, safe_mult=5, U=0.0)
    mc.set_conf(O.random_conf(4, N, mc.slices)); mc.seed(4)
    mc.prepare()
    ut = [UT.UnequalTimeOracle(mc, b) for b in range(2)]
    neighs = O.square_neighs(L)
    rho, c, cd = R.ed_hubbard_greens(neighs, N, 0.0, 1.0, 0.0, beta, return_state=True)
    # rebuild H to evolve operators: H = sum_sigma sum_<src,trg> -t c^dag_trg c_src
    H = np.zeros_like(rho)
    for s_ in range(2):
        for src in range(N):
            for trg in neighs[:, src] - 1:
                H -= cd[N * s_ + trg] @ c[N * s_ + src]
    w, V = np.linalg.eigh(H)

    def heis(A, tau):  # e^{tau H} A e^{-tau H}
        return (V * np.exp(tau * w)) @ V.T @ A @ (V * np.exp(-tau * w)) @ V.T

    G00 = R.full_greens([ut[b].greens(0, 0) for b in range(2)])
    for l in (1, 3, 7, 10):
        tau = l * dtau
        Gl0 = R.full_greens([ut[b].greens(l, 0) for b in range(2)])
        G0l = R.full_greens([ut[b].greens(0, l) for b in range(2)])
        Gll = R.full_greens([ut[b].greens(l, l) for b in range(2)])
        for a_ in range(2 * N):
            for b_ in range(2 * N):
                assert abs(Gl0[a_, b_] - np.trace(rho @ heis(c[a_], tau) @ cd[b_])) < 1e-10
                assert abs(G0l[a_, b_] + np.trace(rho @ heis(cd[b_], tau) @ c[a_])) < 1e-10
        pg = (G00, G0l, Gl0, Gll)
        n_op = [cd[a_] @ c[a_] for a_ in range(2 * N)]
        for i in range(N):
            for j in range(N):
                ni, nj = n_op[i] + n_op[i + N], n_op[j] + n_op[j + N]
                assert abs(R.cdc_kernel_packed(pg, N, i, j) - np.trace(rho @ heis(ni, tau) @ nj)) < 1e-10
                szi, szj = n_op[i] - n_op[i + N], n_op[j] - n_op[j + N]
                assert abs(R.sdc_z_kernel_packed(pg, N, i, j) - np.trace(rho @ heis(szi, tau) @ szj)) < 1e-10
                sxi = cd[i] @ c[i + N] + cd[i + N] @ c[i]
                sxj = cd[j] @ c[j + N] + cd[j + N] @ c[j]
                assert abs(R.sdc_x_kernel_packed(pg, N, i, j) - np.trace(rho @ heis(sxi, tau) @ sxj)) < 1e-10
                syi = cd[i] @ c[i + N] - cd[i + N] @ c[i]      # -i * S_y: the reference skips the factor (measurements.jl:104-106)
                syj = cd[j] @ c[j + N] - cd[j + N] @ c[j]
                assert abs(R.sdc_y_kernel_packed(pg, N, i, j) + np.trace(rho @ heis(syi, tau) @ syj)) < 1e-10
        for s1 in range(N):
            for t1 in range(N):
                for s2 in range(N):
                    for t2 in range(N):
                        ed = np.trace(rho @ heis(c[s1] @ c[N + t1], tau) @ cd[N + t2] @ cd[s2])
                        assert abs(R.pc_kernel_packed(pg, N, s1, t1, s2, t2) - ed) < 1e-10
