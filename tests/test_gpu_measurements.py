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


def test_reference_integration_goldens_on_the_device(gpu):
    """End to end on the GPU against numbers produced by the reference itself
    (test/integration_tests.jl:29-75: attractive 4x4, beta = 1; the published means of one seeded
    Julia run with atol = 4 dtau^2 = 0.04): 32 walkers x 60 measured sweeps, everything - sweeps,
    true Green's function, charge / spin density correlations over EachSitePairByDistance - on the
    device through run().  The golden's own standard error reaches 0.019 (integration_tests.jl:50-52)."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "integration_attractive_4x4.json")))
    model = gpu.HubbardModelAttractive(4, 2)
    mc = gpu.DQMC(model, beta=1.0, n_walkers=32, seed=2024, thermalization=50, sweeps=60, measure_rate=1)
    mc.set_pair_directions(gpu.EachSitePairByDistance(model.l))
    mc.run(measurements=("greens", "correlations"))
    acc = mc.unpack_accumulators(mc.accumulators())
    assert acc["count"] == 32 * 60
    ref = np.array(g["G_mean_colmajor"]).reshape((16, 16), order="F")
    assert np.abs(acc["G"][0] - ref).max() < g["atol"] + 0.02
    c = mc.correlations()
    assert c["count"] == 32 * 60
    assert np.abs(c["CDC"] - np.array(g["CDC_mean"])).max() < g["atol"] + 0.02
    assert np.abs(c["SDCx"] - np.array(g["SDCx_mean"])).max() < g["atol"]
    # structure the reference's result has: on-site first, then four symmetry-equivalent neighbours
    assert c["CDC"][0] > 1.4 and np.ptp(c["CDC"][1:5]) < 0.02
    assert np.abs(c["Mz"]).max() == 0.0
    mc.close()


def test_run_with_all_measurements(gpu):
    model = gpu.HubbardModelRepulsive(4, 2)
    mc = gpu.DQMC(model, beta=1.0, n_walkers=2, seed=5, thermalization=1, sweeps=4, measure_rate=2)
    mc.set_local_targets(gpu.EachLocalQuadByDistance(model.l))
    mc.run(measurements=("greens", "correlations", "pairing", "susceptibilities"))
    assert mc.unpack_accumulators(mc.accumulators())["count"] == 4
    assert mc.correlations()["count"] == 4 and mc.pairing()[1] == 4 and mc.susceptibilities()["count"] == 4
    with pytest.raises(ValueError):
        mc.run(measurements=("nonsense",))
    mc.close()
