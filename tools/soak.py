"""Longer run: acceptance, propagation error and hand-off health over many sweeps."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
m = g.load_package()
for name, model, sweeps in (("attractive", m.HubbardModelAttractive(16, 2), 60), ("repulsive", m.HubbardModelRepulsive(16, 2), 30)):
    mc = m.DQMC(model, beta=8.0, n_walkers=32, seed=77)
    mc.prepare()
    t0 = time.perf_counter()
    mc.sweep(sweeps)
    dt = time.perf_counter() - t0
    a = [mc.analysis(w) for w in range(32)]
    pe = max(x.propagation_error.max if x.propagation_error.count else 0.0 for x in a)
    acc = np.mean([x.acc_rate for x in a])
    g_ = mc.greens_eff(5)[0]
    g0 = mc.calculate_greens(mc.current_slice - 1 if mc.direction == 1 else mc.current_slice, 5)[0] if False else None
    print("%s: %d sweeps in %.2f s (%.1f walker-sweeps/s), acceptance %.4f, max propagation error %.2e, diag(G) in [%.3f, %.3f]" % (
        name, sweeps, dt, 32 * sweeps / dt, acc, pe, np.diag(g_).min(), np.diag(g_).max()), flush=True)
    mc.close()
