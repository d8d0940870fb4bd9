"""Parity at the sizes the bench runs, and of the events the small cases never produce (round 3):
 * BASELINE config 3 with all 32 walkers (256 co-resident cooperative-QR workgroups, 160-workgroup fused sweep launches,
   XCD maps at full occupancy) against 32 oracle chains;
 * config 4's > 64-unit regime (40 repulsive walkers = 80 units: tile-QR hand-over to the tail kernel, split sweep
   launches, two-pass flush) against 40 oracle chains;
 * MagnitudeStats events (DQMC.jl:4-31): propagation errors (stack.jl:538-549, 602-611) and negative determinant
   ratios (DQMC.jl:562-568) made to happen on both sides and compared field by field;
 * the bounded-spin abort of the cooperative QR taken for real (one part stops publishing at step j);
 * Val(false) behind the in-place factorisations with more workgroups than the chip holds at once (the launch order
   dependence behind the wrong G of gpurun_out/r02_t3.log)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import relerr
from test_gpu_dqmc import make_pair, compare, TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cores():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except Exception:
        pass
    return max(1, min(16, os.cpu_count() or 1))


def _oracles(refs, fn):
    """run fn(o) for every oracle chain, one chain per usable host core (ctypes releases the GIL)"""
    with ThreadPoolExecutor(max_workers=_cores()) as ex:
        list(ex.map(fn, refs))


def _counters_equal(mc, refs):
    for w, o in enumerate(refs):
        a, st = mc.analysis(w), o.stats()
        assert (a.prop_local, a.acc_local) == (st.prop_local, st.acc_local), "walker %d" % w
        assert mc.uniforms_used(w) == o.uniforms_used(), "walker %d" % w


def test_cfg3_32_walkers_one_sweep(gpu, O):
    """BASELINE config 3 as benchmarked: attractive 16x16, beta=8 (n=256, M=80), 32 walkers, prepare + 1 sweep"""
    mc, refs = make_pair(gpu, O, 16, "attractive", 8.0, n_walkers=32)
    mc.prepare()
    _oracles(refs, lambda o: o.prepare())
    compare(mc, refs)
    mc.sweep(1)
    _oracles(refs, lambda o: o.sweeps(1))
    compare(mc, refs)
    _counters_equal(mc, refs)
    assert mc.qr_fallbacks() == 0
    for w in range(32):
        a = mc.analysis(w)
        assert a.propagation_error.count == 0 and a.negative_probability.count == 0
    mc.close()


def test_cfg4_80_units_quarter_sweep(gpu, O):
    """BASELINE config 4's per-GPU regime beyond 64 units: repulsive 16x16, beta=8, 40 walkers = 80 units of 256x256
    (single-workgroup tile QR handing over to qr_tail_kernel, elimination and flush as separate launches):
    prepare + 40 updates against the oracle"""
    mc, refs = make_pair(gpu, O, 16, "repulsive", 8.0, n_walkers=40)
    mc.prepare()
    _oracles(refs, lambda o: o.prepare())
    compare(mc, refs)
    for _ in range(40):
        mc.update()

    def adv(o):
        for _ in range(40):
            o.update()
    _oracles(refs, adv)
    compare(mc, refs)
    _counters_equal(mc, refs)
    mc.close()


def _mag_equal(a, b, what):
    assert a.count == b.count, "%s: count %d vs %d" % (what, a.count, b.count)
    if a.count:
        for f in ("max", "min", "sum"):
            x, y = getattr(a, f), getattr(b, f)
            assert abs(x - y) <= 1e-10 * max(1.0, abs(y)), "%s.%s: %r vs %r" % (what, f, x, y)


def _stats_equal(mc, refs):
    for w, o in enumerate(refs):
        a, st = mc.analysis(w), o.stats()
        assert (a.prop_local, a.acc_local) == (st.prop_local, st.acc_local)
        _mag_equal(a.propagation_error, st.propagation_error, "walker %d propagation_error" % w)
        _mag_equal(a.negative_probability, st.negative_probability, "walker %d negative_probability" % w)
        assert a.imaginary_probability.count == 0 and st.imaginary_probability.count == 0


def _inject(mc, refs, make):
    """the same effective Green's function on both sides: the oracle's current one, modified by make(walker, blocks)"""
    for w, o in enumerate(refs):
        blocks = make(w, [b.copy() for b in o.greens_eff()])
        o.set_greens_eff(blocks)
        mc.set_greens_eff(w, blocks)


@pytest.mark.parametrize("kind,L,beta", [("attractive", 4, 2.0), ("repulsive", 4, 2.0), ("repulsive", 16, 2.0)])
def test_magnitude_stats_events(gpu, O, kind, L, beta):
    """DQMCAnalysis.propagation_error / negative_probability with events in them: count exact, log10 max / min / sum
    within 1e-10 of the oracle.  Propagation errors come from a perturbation of mc.s.greens injected on both sides
    (the wrapped G then differs from the recomputed one by the propagated perturbation at the next stabilisation,
    in either direction); negative determinant ratios (repulsive model: R_up R_dn < 0, Repulsive.jl:128-156) from
    diagonal entries outside [0, 1] injected before a sweep_spatial."""
    nw = 3 if L == 4 else 2
    mc, refs = make_pair(gpu, O, L, kind, beta, n_walkers=nw)
    n = L * L
    mc.prepare()
    _oracles(refs, lambda o: o.prepare())
    rng = np.random.default_rng(17)
    noise = [[rng.standard_normal((n, n)) for _ in range(mc.nb)] for _ in range(nw)]

    def both(k):
        for _ in range(k):
            mc.update()

        def adv(o):
            for _ in range(k):
                o.update()
        _oracles(refs, adv)

    both(3)
    # (1) perturbation on the way down: fires at the next stabilisation of the down pass (stack.jl:596-612)
    _inject(mc, refs, lambda w, bl: [b + 0.05 / (1 + w) * noise[w][i] for i, b in enumerate(bl)])
    both(mc.p.safe_mult + 1)
    _stats_equal(mc, refs)
    assert all(mc.analysis(w).propagation_error.count >= 1 for w in range(nw))
    # (2) out-of-range diagonal: negative determinant ratios in the repulsive model (DQMC.jl:562-568)
    sites = list(range(1, n, max(1, n // 12)))

    def diag(w, bl):
        for i in sites[0::2]:
            bl[0][i, i] = 3.0 + 0.1 * w
        for i in sites[1::2]:
            bl[-1][i, i] = -1.5 - 0.1 * w
        return bl
    _inject(mc, refs, diag)
    mc.sweep_spatial()
    _oracles(refs, lambda o: o.sweep_spatial())
    _stats_equal(mc, refs)
    for w, o in enumerate(refs):
        assert np.array_equal(mc.conf(w), o.conf())
    if kind == "repulsive":
        assert sum(mc.analysis(w).negative_probability.count for w in range(nw)) >= 2
    else:  # the attractive ratio is a square (Attractive.jl:113-127): never negative
        assert all(mc.analysis(w).negative_probability.count == 0 for w in range(nw))
    # (3) through the lower turn-around - which recomputes G without a check (stack.jl:506-517), so the damage of (2)
    # leaves no event of its own there - and up to the first stabilisation of the up pass (stack.jl:519-550), which sees a
    # second perturbation (through the wrap of the old G, stack.jl:534-536)
    both(mc.p.safe_mult)
    _inject(mc, refs, lambda w, bl: [b + 1e-3 * noise[w][i].T for i, b in enumerate(bl)])
    both(2 * mc.p.slices - 3 - 2 * mc.p.safe_mult - 1)
    _stats_equal(mc, refs)
    assert all(mc.analysis(w).propagation_error.count >= 2 for w in range(nw))
    for w, o in enumerate(refs):
        assert np.array_equal(mc.conf(w), o.conf())
    compare(mc, refs, tol=1e-9, conf=False)
    mc.close()


@pytest.mark.parametrize("n,step,env", [(256, 5, {"DQMC_QR_NOBLOCKED": "1"}), (256, 100, {"DQMC_QR_NOBLOCKED": "1"}), (64, 17, {})])
def test_cooperative_qr_dropout_is_recovered(gpu, O, n, step, env):
    """DQMC_QR_FORCE_TIMEOUT=step:<j>: one of the eight workgroups of every matrix stops publishing at step j.  The
    others run into the bounded spins (a genuine abort: partially written output, tau, pivots), raise the fallback
    word, the tail kernel stands down and the guarded single-workgroup kernel redoes the factorisation from the
    untouched input.  One launch (the spins take ~0.2 s each).  (n = 256: the pivoted cooperative QR behind
    DQMC_QR_NOBLOCKED; the default one-launch UDT reports the time-out instead: next test.)"""
    rng = np.random.default_rng(n + step)
    X = rng.standard_normal((2, n, n))
    X[1] *= np.exp(rng.uniform(-10, 10, size=n))[None, :]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["DQMC_QR_FORCE_TIMEOUT"] = "step:%d" % step
    try:
        U, D, T, piv = gpu.udt_AVX_pivot(X, False)
        del os.environ["DQMC_QR_FORCE_TIMEOUT"]
        U0, D0, T0, piv0 = gpu.udt_AVX_pivot(X, False)
    finally:
        os.environ.pop("DQMC_QR_FORCE_TIMEOUT", None)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for i in range(2):
        P = np.zeros((n, n)); P[np.arange(n), piv[i] - 1] = 1
        rec = (U[i] * D[i]) @ np.triu(T[i]) @ P
        scale = np.abs(X[i]).max(axis=0)
        assert (np.abs(rec - X[i]) / scale[None, :]).max() < 1e-12
        assert np.array_equal(piv[i], piv0[i])
        assert relerr(D[i], D0[i]) < 1e-12


@pytest.mark.parametrize("step", [3, 20])
def test_blocked_udt_dropout_is_reported(gpu, step):
    """The one-launch UDT (csrc/qrb.hip) under the same test hook: part 3 of every matrix stops publishing at step j of its
    panel, the parts behind it run out of their bounded spins, every workgroup still leaves through its normal path (no
    hang), the device error word is raised and the CALL FAILS with a message that names the cause - its results are not
    returned as if they were valid.  (A workgroup only ever waits for lower parts of its own matrix, which in-order
    dispatch has started before it: the wait cannot fail on a healthy device, so there is no silent second path here.)
    The next call on the same device is unaffected."""
    rng = np.random.default_rng(step)
    X = rng.standard_normal((2, 256, 256))
    os.environ["DQMC_QR_FORCE_TIMEOUT"] = "step:%d" % step
    try:
        with pytest.raises(gpu.DQMCError, match="timed out"):
            gpu.udt_AVX_pivot(X, True)
    finally:
        del os.environ["DQMC_QR_FORCE_TIMEOUT"]
    U, D, T, piv = gpu.udt_AVX_pivot(X, True)
    for i in range(2):
        assert relerr((U[i] * D[i]) @ T[i], X[i]) < 1e-12


def test_dropout_through_the_engine_counts_fallbacks(gpu, O):
    """the same abort inside a handle: results stay in parity with the oracle and dqmc_qr_fallbacks counts the launches"""
    os.environ["DQMC_QR_FORCE_TIMEOUT"] = "step:3"
    try:
        mc, refs = make_pair(gpu, O, 4, "attractive", 1.0, n_walkers=2)
    finally:
        del os.environ["DQMC_QR_FORCE_TIMEOUT"]
    mc.prepare()
    for o in refs:
        o.prepare()
    compare(mc, refs)
    assert mc.qr_fallbacks() == 3  # build_stack: 1 UDT, calculate_greens: 2
    mc.close()


@pytest.mark.parametrize("n,batch", [(64, 700), (256, 300)])
def test_val_false_behind_in_place_factorisations(gpu, n, batch):
    """udt_AVX_pivot!(..., Val(false)) (UDT.jl:298-306) rescales the factored matrix in place.  Behind the in-place
    factorisations (DQMC_QR_NOCOOP here; in production: more than 64 units, n > 256) udt_finish runs 8 workgroups per
    matrix on the SAME buffer it reads D from, with more workgroups than the chip holds at once: every row of every
    T must still carry 1/D of the factorisation (T has a unit diagonal up to sign, U D triu(T) P = X)."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((batch, n, n)) * np.exp(rng.uniform(-3, 3, size=(batch, 1, n)))
    os.environ["DQMC_QR_NOCOOP"] = "1"
    try:
        U, D, T, piv = gpu.udt_AVX_pivot(X, False)
    finally:
        del os.environ["DQMC_QR_NOCOOP"]
    dg = np.abs(np.einsum("bii->bi", T))
    assert np.abs(dg - 1.0).max() < 1e-12
    for i in range(0, batch, max(1, batch // 40)):
        P = np.zeros((n, n)); P[np.arange(n), piv[i] - 1] = 1
        rec = (U[i] * D[i]) @ np.triu(T[i]) @ P
        scale = np.abs(X[i]).max(axis=0)
        assert (np.abs(rec - X[i]) / scale[None, :]).max() < 1e-12


@pytest.mark.parametrize("kind,L,beta", [("attractive", 8, 2.0), ("repulsive", 16, 2.0)])
def test_checks_switched_off(gpu, O, kind, L, beta):
    """check_propagation_error = check_sign_problem = false (DQMCParameters, DQMC.jl:77-78): the stabilisation steps
    skip the wrap of the old Green's function and the comparison (stack.jl:534-549, 596-612), sweep_spatial the sign
    bookkeeping (DQMC.jl:554-569) - same trajectory as the oracle with the same switches, no events recorded"""
    mc, refs = make_pair(gpu, O, L, kind, beta, n_walkers=2, check_propagation_error=False, check_sign_problem=False)
    refs = []
    for w in range(2):
        mk = {"mu": mc.model.mu} if kind == "attractive" else {}
        o = O.OracleDQMC(L, kind, beta=beta, delta_tau=mc.p.delta_tau, safe_mult=mc.p.safe_mult, U=mc.model.U,
                         check_propagation_error=False, check_sign_problem=False, **mk)
        o.set_conf(mc.conf(w)); o.seed(mc.seeds[w])
        refs.append(o)
    mc.prepare()
    _oracles(refs, lambda o: o.prepare())
    compare(mc, refs)
    mc.sweep(1)
    _oracles(refs, lambda o: o.sweeps(1))
    compare(mc, refs)
    _counters_equal(mc, refs)
    for w in range(2):
        a = mc.analysis(w)
        assert a.propagation_error.count == 0 and a.negative_probability.count == 0
    mc.close()
