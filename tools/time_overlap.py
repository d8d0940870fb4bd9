"""config 3: ms per sweep with the auxiliary-stream overlap (default), without the look-ahead chain products
(DQMC_NO_CHAIN_AHEAD) and on one stream (DQMC_NO_OVERLAP); same seeds, final HS fields compared"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
m = g.load_package()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ref = None
for name, env in (("one_stream", {}), ("overlap", {"DQMC_OVERLAP": "1"}), ("+rdivp", {"DQMC_OVERLAP": "1", "DQMC_OVERLAP_RDIVP": "1"}),
                  ("+chain_ahead", {"DQMC_OVERLAP": "1", "DQMC_CHAIN_AHEAD": "1"})):
    os.environ.update(env)
    mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=W)
    for k in env: del os.environ[k]
    mc.prepare(); mc.sweep(2)
    t0 = time.perf_counter(); mc.sweep(NS); dt = (time.perf_counter() - t0) / NS
    mc.timing_enable(True); mc.sweep(2); tm = mc.timing(); mc.timing_enable(False)
    conf = [mc.conf(w).copy() for w in range(W)]
    gg = mc.greens_eff(0)[0].copy()
    if ref is None: ref = (conf, gg)
    same = all(np.array_equal(a, b) for a, b in zip(conf, ref[0]))
    print("%-15s %.2f ms/sweep -> %.1f w-s/s | conf same as first: %s, G relerr %.2e | " % (name, dt * 1e3, W / dt, same, np.abs(gg - ref[1]).max() / np.abs(ref[1]).max())
          + "  ".join("%s %.1f" % (f, v[0] / 2) for f, v in tm.items()) + " | fallbacks %d" % mc.qr_fallbacks(), flush=True)
    mc.close()
