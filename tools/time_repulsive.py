import sys, os, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mc = m.DQMC(m.HubbardModelRepulsive(16, 2), beta=8.0, n_walkers=W, seed=1)
mc.prepare(); mc.sweep(1)
t0 = time.perf_counter(); mc.sweep(3); dt = (time.perf_counter() - t0) / 3
mc.timing_enable(True); mc.sweep(1); t = mc.timing()
print("config 4 (repulsive 16x16, beta=8), %d walkers: %.1f ms/sweep -> %.1f walker-sweeps/s; device ms: %s" % (
    W, dt * 1e3, W / dt, {k: round(v[0], 1) for k, v in t.items()}))
