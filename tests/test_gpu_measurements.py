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


def _device_blocks(gpu, kind, L, K, n_walkers, blocks, per_block, therm, seed):
    """block means of every observable of the reference's integration testsets, measured on the device (true Green's
    function, correlations over EachSitePairByDistance, pairing over EachLocalQuadByDistance{K}, the HS field)"""
    model = (gpu.HubbardModelAttractive if kind == "attractive" else gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=1.0, n_walkers=n_walkers, seed=seed)
    mc.set_local_targets(gpu.EachLocalQuadByDistance(model.l, K))
    mc.prepare()
    for _ in range(therm):
        mc.update_until_measure()
    mc.reset_accumulators()
    N, nb = L * L, mc.nb
    ser = {k: [] for k in ("G", "conf", "CDC", "SDCx", "SDCy", "SDCz", "Mx", "My", "Mz", "PC")}
    prev_acc, prev_corr, prev_pc = np.zeros(mc.accumulator_size()), None, None
    nd = None
    for _ in range(blocks):
        for _ in range(per_block):
            mc.update_until_measure()
            mc.accumulate_greens(); mc.accumulate_correlations(); mc.accumulate_pairing()
        acc, corr = mc.accumulators(), mc.correlations_raw()
        pc_mean, pc_cnt = mc.pairing()
        pc = pc_mean * pc_cnt
        if prev_corr is None:
            prev_corr, prev_pc = np.zeros_like(corr), np.zeros_like(pc)
            nd = (len(corr) - 1 - 3 * N) // 4
        cnt = acc[-1] - prev_acc[-1]
        dG = (acc - prev_acc)[:nb * N * N] / cnt
        ser["G"].append(np.stack([dG[b * N * N:(b + 1) * N * N].reshape((N, N), order="F") for b in range(nb)]))
        dc = (corr - prev_corr) / (corr[-1] - prev_corr[-1])
        for i, k in enumerate(("CDC", "SDCx", "SDCy", "SDCz")):
            ser[k].append(dc[i * nd:(i + 1) * nd])
        for i, k in enumerate(("Mx", "My", "Mz")):
            ser[k].append(dc[4 * nd + i * N:4 * nd + (i + 1) * N])
        ser["PC"].append((pc - prev_pc) / per_block / n_walkers)
        ser["conf"].append(np.mean([mc.conf(w).astype(float) for w in range(n_walkers)], axis=0))
        prev_acc, prev_corr, prev_pc = acc, corr, pc
    mc.close()
    return {k: np.array(v) for k, v in ser.items()}


def _block_error(x):
    return x.std(axis=0, ddof=1) / np.sqrt(x.shape[0])


def test_reference_integration_goldens_on_the_device(gpu):
    """End to end on the GPU against EVERY number the reference holds for its two seeded DQMC integration runs
    (test/integration_tests.jl:29-94 attractive 4x4, :98-185 repulsive 2x2): mean G, mean HS field, CDC, SDCx/y/z,
    Mx/y/z, PC.  32 walkers; rules of tests/golden_stats.py (the reference's own atol, unwidened, wherever its own
    published std_error allows it - everywhere for CDC / SDC / PC / M - and a z-score against the published std_error
    for every element)."""
    import golden_stats as gs
    g = gs.load("integration_attractive_4x4.json")
    A, atol = g["all"], g["atol"]
    s = _device_blocks(gpu, "attractive", 4, 5, 32, 24, 8, 30, 2024)
    m, se = gs.golden_arrays(A["G"], (16, 16))
    r = gs.check("att G", s["G"][:, 0].mean(0), _block_error(s["G"][:, 0]), m, se, atol)
    assert r["n_at_ref_atol"] >= 200
    m, _ = gs.golden_arrays(A["conf"])
    gs.check("att conf", s["conf"].mean(0), _block_error(s["conf"]), m,
             np.sqrt(np.maximum(1 - m ** 2, 0.0) / g["n_measurements"]), atol)
    for k in ("CDC", "SDCx", "SDCy", "SDCz"):
        m, se = gs.golden_arrays(A[k])
        r = gs.check("att " + k, s[k].mean(0), _block_error(s[k]), m, se, atol, all_at_atol=True)
        assert r["n_at_ref_atol"] == 16
    m, se = gs.golden_arrays(A["PC"], (16, 5, 5))
    r = gs.check("att PC", s["PC"].mean(0).reshape((16, 5, 5), order="F"),
                 _block_error(s["PC"]).reshape((16, 5, 5), order="F"), m, se, atol, all_at_atol=True, zmax=6.5)
    assert r["n_at_ref_atol"] == 400
    assert np.abs(s["Mz"]).max() == 0.0  # attractive: identical spin blocks
    # structure the reference's result has: on-site first, then four symmetry-equivalent neighbours
    cdc = s["CDC"].mean(0)
    assert cdc[0] > 1.4 and np.ptp(cdc[1:5]) < 0.02

    g = gs.load("integration_repulsive_2x2.json")
    A, atol = g["all"], g["atol"]
    s = _device_blocks(gpu, "repulsive", 2, 3, 32, 24, 8, 30, 2025)
    m, se = gs.golden_arrays(A["G"])
    ours = np.zeros((8, 8)); ours_se = np.zeros((8, 8))
    for b in range(2):
        ours[4 * b:4 * b + 4, 4 * b:4 * b + 4] = s["G"][:, b].mean(0)
        ours_se[4 * b:4 * b + 4, 4 * b:4 * b + 4] = _block_error(s["G"][:, b])
    gs.check("rep G", ours, ours_se, m, se, atol)
    m, _ = gs.golden_arrays(A["conf"])
    gs.check("rep conf", s["conf"].mean(0), _block_error(s["conf"]), m,
             np.sqrt(np.maximum(1 - m ** 2, 0.0) / g["n_measurements"]), atol)
    for k in ("CDC", "Mx", "My", "Mz", "SDCx", "SDCy", "SDCz"):
        m, se = gs.golden_arrays(A[k])
        gs.check("rep " + k, s[k].mean(0), _block_error(s[k]), m, se, atol if A[k]["has_atol"] else None,
                 all_at_atol=True)
    m, se = gs.golden_arrays(A["PC"], (4, 3, 3))
    gs.check("rep PC", s["PC"].mean(0).reshape((4, 3, 3), order="F"),
             _block_error(s["PC"]).reshape((4, 3, 3), order="F"), m, se, None)


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
