"""BASELINE configurations at their FULL sizes through size-independent properties (the CPU oracle
needs minutes per sweep there).  Config 5: 24x24, beta=20, dtau=0.05 (n=576, 400 slices): the deep
UDT-stabilisation stress case."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cfg5_full_depth_properties(gpu):
    model = gpu.HubbardModelAttractive(24, 2)
    mc = gpu.DQMC(model, beta=20.0, delta_tau=0.05, n_walkers=2, seed=11, check_propagation_error=True)
    M = mc.p.slices
    assert (mc.N, M, mc.p.safe_mult) == (576, 400, 10)
    t0 = time.time()
    mc.prepare()
    mc.update_until_measure()          # the down-sweep that follows build_stack (DQMC.jl:412-414)
    mc.sweep(1)                        # one full sweep: up and down, every slice visited twice
    dt = time.time() - t0
    assert mc.current_slice == 1 and mc.direction == 1
    for w in range(2):
        a = mc.analysis(w)
        assert a.prop_local == 3 * 576 * M and 0.2 < a.acc_rate < 1.0
        # stack.jl:538-549: |G_wrapped - G_recomputed| at every stabilisation point of the sweep
        assert a.propagation_error.count == 0 or a.propagation_error.max < 1e-6
    # the propagated Green's function equals the from-scratch one of the same configuration
    g_prop = [mc.greens_eff(w)[0] for w in range(2)]
    for w in range(2):
        g0 = mc.calculate_greens(0, w)[0]
        assert np.abs(g_prop[w] - g0).max() < 1e-9
        assert np.abs(np.diag(g0)).max() <= 1.0 + 1e-9 and np.abs(np.diag(g0)).min() >= -1e-9  # occupations
    # configuration wire format: compress -> decompress -> replay reproduces G (HubbardModel.jl:56-59, DQMC.jl:647-653)
    bits = [mc.conf_bits(w) for w in range(2)]
    confs = [mc.conf(w) for w in range(2)]
    mc2 = gpu.DQMC(model, beta=20.0, delta_tau=0.05, n_walkers=2, seed=99)
    for w in range(2):
        mc2.set_conf_bits(w, bits[w])
        assert np.array_equal(mc2.conf(w), confs[w])
    mc2.replay_greens(0)
    for w in range(2):
        assert np.abs(mc2.greens_eff(w)[0] - g_prop[w]).max() < 1e-9
    mc2.close()
    # unequal-time identities at full depth (flavortests_DQMC.jl:107-118)
    for k in (0, 137, M):
        assert np.abs(mc.calculate_greens(k, 1)[0] - mc.calculate_greens_kl(k, k, 1)[0]).max() < 1e-10
    for k in (1, 250):
        assert np.abs(mc.greens_kl(k, 0, 0)[0] + mc.greens_kl(k, M, 0)[0]).max() < 1e-10
    print("cfg5 full size: prepare + 1 sweep of 2 walkers in %.1f s" % dt)
    mc.close()


def test_cfg4_full_depth_properties(gpu):
    """Config 4: repulsive 16x16, beta=8 (two 256x256 blocks sharing one HS field), a full sweep"""
    model = gpu.HubbardModelRepulsive(16, 2)
    mc = gpu.DQMC(model, beta=8.0, n_walkers=4, seed=23)
    M = mc.p.slices
    mc.prepare()
    mc.update_until_measure()
    mc.sweep(1)
    assert mc.current_slice == 1 and mc.direction == 1
    for w in range(4):
        a = mc.analysis(w)
        assert a.prop_local == 3 * 256 * M and 0.2 < a.acc_rate < 1.0
        assert a.propagation_error.count == 0 or a.propagation_error.max < 1e-6
        assert a.negative_probability.count == 0          # half filling: no sign problem (Repulsive.jl:128-156)
        g = mc.greens_eff(w)
        g0 = mc.calculate_greens(0, w)
        for b in range(2):
            assert np.abs(g[b] - g0[b]).max() < 1e-9
    # particle-hole symmetry of the half-filled bipartite model, configuration by configuration:
    # G_dn[i,j] = delta_ij - (-1)^(i+j) G_up[j,i] for the true Green's function
    L = 16
    sgn = np.array([(-1) ** ((i % L) + (i // L)) for i in range(L * L)], dtype=float)
    for w in range(4):
        up, dn = mc.greens(w)
        assert np.abs(dn - (np.eye(L * L) - (sgn[:, None] * sgn[None, :]) * up.T)).max() < 1e-9
    for k in (0, 33, M):
        a_, b_ = mc.calculate_greens(k, 2), mc.calculate_greens_kl(k, k, 2)
        assert max(np.abs(a_[b] - b_[b]).max() for b in range(2)) < 1e-10
    mc.close()
