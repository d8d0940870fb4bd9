"""equal-time correlation measurements on the device against the generic 2N x 2N
kernel formulas restated in oracle/ref_test_oracle.py (measurements.jl:51-190)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_correlations_match_reference_formulas(gpu, O, R, kind):
    L = 4
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=1.0, n_walkers=3, seed=31)
    it = gpu.EachSitePairByDistance(model.l)
    mc.set_pair_directions(it)
    mc.prepare()
    mc.update_until_measure()
    mc.reset_accumulators()
    mc.accumulate_correlations()
    mc.update_until_measure()
    mc.accumulate_correlations()
    res = mc.correlations()
    assert res["count"] == 6
    # reference: same trajectories with the oracle, formulas in their generic form
    ref = None
    for w in range(3):
        o = O.OracleDQMC(L, kind, beta=1.0)
        rng = np.random.Generator(np.random.Philox(key=mc.seeds[w]))
        o.set_conf(gpu.rand_conf(rng, 16, 10)); o.seed(mc.seeds[w])
        o.prepare()
        for _ in range(2):
            o.update_until_measure()
            c = R.equal_time_correlations(o.greens(), L, kind == "attractive")
            ref = c if ref is None else {k: ref[k] + c[k] for k in c}
    for k in ("CDC", "SDCx", "SDCy", "SDCz", "Mx", "My", "Mz"):
        assert np.abs(res[k] - ref[k] / 6).max() < 1e-10, k
    mc.close()


@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_pairing_correlation(gpu, O, R, kind):
    """pairing_correlation = pc_kernel over EachLocalQuadByDistance{5} (measurements.jl:199-214)"""
    L = 4
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=1.0, n_walkers=3, seed=47)
    it = gpu.EachLocalQuadByDistance(model.l)
    mc.set_local_targets(it)
    mc.prepare()
    mc.reset_accumulators()
    ref = np.zeros((16, 5, 5))
    for _ in range(2):
        mc.update_until_measure()
        mc.accumulate_pairing()
        for w in range(3):
            ref += R.pairing_correlation(mc.greens(w), L, kind == "attractive", 5)
    out, cnt = mc.pairing()
    assert cnt == 6 and out.shape == (16, 5, 5)
    assert np.abs(out - ref / 6).max() < 1e-12
    # the same sums through the host iterator in the reference's iteration order
    G = R.full_greens(mc.greens(0))
    lin_sum = np.zeros(16 * 25)
    for lin, s1, t1, s2, t2 in it:
        lin_sum[lin - 1] += R.pc_kernel(G, 16, s1 - 1, t1 - 1, s2 - 1, t2 - 1)
    one = R.pairing_correlation(mc.greens(0), L, kind == "attractive", 5)
    assert np.abs(lin_sum.reshape((16, 5, 5), order="F") / 16 - one).max() < 1e-12
    mc.reset_accumulators()
    mc.accumulate_pairing()
    assert mc.pairing()[1] == 3
    mc.close()


def test_pairing_requires_tables(gpu):
    mc = gpu.DQMC(gpu.HubbardModelAttractive(4, 2), beta=1.0, n_walkers=1)
    mc.prepare()
    with pytest.raises(RuntimeError):
        mc.accumulate_pairing()
    mc.close()
