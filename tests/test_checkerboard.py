"""the checkerboard (CheckerboardTrue) propagator.  The reference applies one sparse
matrix per bond group (stack.jl:185-235, slice_matrices.jl:79-222); here the group products are
multiplied out on the host and run through the dense MFMA path.  Mirrors test/slice_matrices.jl:85-129."""
import numpy as np
import pytest
from scipy.linalg import expm


def _group_matrices(mc_amd, model, dtau):
    """chkr_hop_half etc. rebuilt independently of checkerboard_exponentials' products"""
    T = model.hopping_matrix()[0]
    N = T.shape[0]
    cb, groups, n = mc_amd.build_checkerboard(model.l)
    H, Hinv, C, Cinv = [], [], [], []
    for gs, ge in groups:
        Tg = np.zeros((N, N))
        sites = []
        for i in range(gs, ge + 1):
            src, trg = cb[0, i - 1], cb[1, i - 1]
            Tg[trg - 1, src - 1] = T[trg - 1, src - 1]
            sites += [src, trg]
        assert len(set(sites)) == len(sites)  # bonds of a group share no site (abstract.jl:23-54)
        H.append(expm(-0.5 * dtau * Tg)); Hinv.append(expm(0.5 * dtau * Tg))
        C.append(expm(-dtau * Tg)); Cinv.append(expm(dtau * Tg))
    mus = np.diag(T)
    return H, Hinv, C, Cinv, np.diag(np.exp(-dtau * mus)), np.diag(np.exp(dtau * mus)), n


def test_checkerboard_constants_follow_the_reference_sequences(mc_amd):
    dtau = 0.1
    model = mc_amd.HubbardModelAttractive(4, 2, mu=0.4)
    T = model.hopping_matrix()[0]
    N = 16
    eT, eTinv, eT2, eTinv2 = mc_amd.checkerboard_exponentials(T, model.l, dtau)
    H, Hinv, C, Cinv, Mu, Muinv, n = _group_matrices(mc_amd, model, dtau)
    rng = np.random.default_rng(3)
    conf = rng.choice([-1.0, 1.0], N)
    lam = np.arccosh(np.exp(0.5 * dtau * 1.0))
    eV, eVinv = np.diag(np.exp(lam * conf)), np.diag(np.exp(-lam * conf))
    M0 = rng.standard_normal((N, N))

    def left(M):   # multiply_slice_matrix_left! (slice_matrices.jl:104-124)
        M = Mu @ (eV @ M)
        for i in reversed(range(1, n)): M = H[i] @ M
        M = C[0] @ M
        for i in range(1, n): M = H[i] @ M
        return M

    def right(M):  # multiply_slice_matrix_right! (:125-149)
        for i in reversed(range(1, n)): M = M @ H[i]
        M = M @ C[0]
        for i in range(1, n): M = M @ H[i]
        return (M @ Mu) @ eV

    def inv_left(M):  # :150-171
        for i in reversed(range(1, n)): M = Hinv[i] @ M
        M = Cinv[0] @ M
        for i in range(1, n): M = Hinv[i] @ M
        return eVinv @ (Muinv @ M)

    def inv_right(M):  # :172-194
        M = (M @ eVinv) @ Muinv
        for i in reversed(range(1, n)): M = M @ Hinv[i]
        M = M @ Cinv[0]
        for i in range(1, n): M = M @ Hinv[i]
        return M

    def dagger_left(M):  # :195-222
        for i in reversed(range(1, n)): M = H[i].T @ M
        M = C[0].T @ M
        for i in range(1, n): M = H[i].T @ M
        return eV @ (Mu @ M)

    B, Binv = eT2 @ eV, eVinv @ eTinv2
    tol = 1e-13
    assert np.abs(left(M0) - B @ M0).max() < tol
    assert np.abs(right(M0) - M0 @ B).max() < tol
    assert np.abs(inv_left(M0) - Binv @ M0).max() < tol
    assert np.abs(inv_right(M0) - M0 @ Binv).max() < tol
    assert np.abs(dagger_left(M0) - B.T @ M0).max() < tol
    assert np.abs(B @ Binv - np.eye(N)).max() < 1e-13
    # _greens! (DQMC.jl:731-750)
    G = M0.copy()
    for i in reversed(range(n)): G = G @ H[i]
    for i in reversed(range(n)): G = Hinv[i] @ G
    assert np.abs(G - eTinv @ M0 @ eT).max() < tol
    # test/slice_matrices.jl:93-129: within 2 dtau of the dense slice matrix and its inverse
    d = mc_amd.hopping_exponentials(T, dtau)
    dense = d[0] @ d[0] @ eV
    assert np.abs(B - dense).max() < 2 * dtau
    assert np.abs(Binv - np.linalg.inv(dense)).max() < 2 * dtau
    assert np.abs(left(M0 / 4) - dense @ (M0 / 4)).max() < 2 * dtau


def test_checkerboard_chain_on_the_oracle(O, mc_amd):
    """the checkerboard run is a DQMC run with other constants: same machinery, Green's function
    within O(dtau) of the dense one on the same configuration"""
    model = mc_amd.HubbardModelAttractive(4, 2)
    T = model.hopping_matrix()[0]
    ex = mc_amd.checkerboard_exponentials(T, model.l, 0.1)
    a = O.OracleDQMC(4, "attractive", beta=1.0, exps=ex)
    b = O.OracleDQMC(4, "attractive", beta=1.0)
    conf = O.random_conf(8, 16, 10)
    for o in (a, b):
        o.set_conf(conf); o.seed(8); o.prepare()
    ga, gb = a.greens_eff()[0], b.greens_eff()[0]
    # (this version of the reference puts only T[trg, src] of each bond into the group matrices, so the
    # two decompositions differ at O(dtau t) per slice, not O(dtau^2): a loose sanity bound only)
    assert 1e-6 < np.abs(ga - gb).max() < 1.0
    assert np.abs(a.calculate_greens_at(3)[0] - np.linalg.inv(np.eye(16) + np.linalg.multi_dot(
        [a.slice_matrix(l)[0] for l in (3, 2, 1, 10, 9, 8, 7, 6, 5, 4)]))).max() < 1e-10
    a.sweeps(2)
    assert a.stats().acc_local > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_checkerboard_on_device(gpu, O, kind):
    L = 4
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=2.0, n_walkers=2, seed=17, checkerboard="sparse")
    ex = gpu.checkerboard_exponentials(model.hopping_matrix()[0], model.l, 0.1)
    refs = []
    for w in range(2):
        o = O.OracleDQMC(L, kind, beta=2.0, exps=ex)
        o.set_conf(mc.conf(w)); o.seed(mc.seeds[w])
        refs.append(o)
    mc.prepare()
    for o in refs:
        o.prepare()
    mc.sweep(2)
    for w, o in enumerate(refs):
        o.sweeps(2)
        assert np.array_equal(mc.conf(w), o.conf())
        for b in range(mc.nb):
            r = o.greens_eff()[b]
            assert np.abs(mc.greens_eff(w)[b] - r).max() < 1e-10 * max(1.0, np.abs(r).max())
            r = o.greens()[b]
            assert np.abs(mc.greens(w)[b] - r).max() < 1e-10 * max(1.0, np.abs(r).max())
    # the dense and the checkerboard chains are different Trotter decompositions of the same model
    md = gpu.DQMC(model, beta=2.0, n_walkers=2, seed=17)
    md.prepare()
    mc2 = gpu.DQMC(model, beta=2.0, n_walkers=2, seed=17, checkerboard=True)
    mc2.prepare()
    d = np.abs(md.greens_eff(0)[0] - mc2.greens_eff(0)[0]).max()
    assert 1e-6 < d < 1.0
    for x in (mc, md, mc2):
        x.close()


def test_sparse_factor_tables_reproduce_the_dense_constants(mc_amd):
    """checkerboard_tables (ELL factors + sequences handed to dqmc_set_checkerboard) against the multiplied-out
    constants: applying sequence q factor by factor equals the dense product"""
    dtau = 0.1
    for model in (mc_amd.HubbardModelAttractive(4, 2, mu=0.3), mc_amd.HubbardModelRepulsive(6, 2)):
        T = model.hopping_matrix()[0]
        N = T.shape[0]
        tb = mc_amd.checkerboard_tables(T, model.l, dtau)
        eT, eTinv, eT2, eTinv2 = mc_amd.checkerboard_exponentials(T, model.l, dtau)
        def dense(i):
            m = np.zeros((N, N))
            for r in range(N):
                for j in range(tb["kmax"]):
                    m[r, tb["cols"][i, r, j]] += tb["vals"][i, r, j]
            return m
        def left(seq):
            M = np.eye(N)
            for i in seq:
                M = dense(i) @ M
            return M
        def right(seq):   # columns mixed by the rows of the stored (transposed) factors
            M = np.eye(N)
            for i in seq:
                M = M @ dense(i).T
            return M
        Mu, Mui = np.diag(tb["mu"]), np.diag(tb["mu_inv"])
        s = tb["seqs"]
        assert np.abs(left(s[0]) @ Mu - eT2).max() < 1e-13          # B / eV
        assert np.abs(Mui @ left(s[1]) - eTinv2).max() < 1e-13      # eV B^-1
        assert np.abs(Mu @ left(s[2]) - eT2.T).max() < 1e-13        # B' / eV
        assert np.abs(right(s[3]) @ Mu - eT2).max() < 1e-13         # X B
        assert np.abs(Mui @ right(s[4]) - eTinv2).max() < 1e-13     # X B^-1
        assert np.abs(right(s[5]) - eT).max() < 1e-13
        assert np.abs(left(s[6]) - eTinv).max() < 1e-13
        assert tb["kmax"] <= 4


@pytest.mark.gpu
@pytest.mark.parametrize("kind,L", [("attractive", 4), ("repulsive", 4), ("attractive", 8)])
def test_sparse_and_dense_checkerboard_paths_agree(gpu, kind, L):
    """the sparse-factor kernel (cb.hip) against the same decomposition run through the dense MFMA path"""
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    a = gpu.DQMC(model, beta=2.0, n_walkers=2, seed=3, checkerboard="sparse")
    b = gpu.DQMC(model, beta=2.0, n_walkers=2, seed=3, checkerboard="dense")
    a.prepare(); b.prepare()
    for _ in range(3):
        for w in range(2):
            for blk in range(a.nb):
                ga, gb = a.greens_eff(w)[blk], b.greens_eff(w)[blk]
                assert np.abs(ga - gb).max() < 1e-11 * max(1.0, np.abs(gb).max())
                ga, gb = a.greens(w)[blk], b.greens(w)[blk]
                assert np.abs(ga - gb).max() < 1e-11 * max(1.0, np.abs(gb).max())
        a.sweep(1); b.sweep(1)
        for w in range(2):
            assert np.array_equal(a.conf(w), b.conf(w))
    ga = a.calculate_greens(7, 0)[0]
    gb = b.calculate_greens(7, 0)[0]
    assert np.abs(ga - gb).max() < 1e-11 * max(1.0, np.abs(gb).max())
    a.close(); b.close()
