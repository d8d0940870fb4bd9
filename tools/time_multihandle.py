import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
W = 32
for groups in (1, 2, 4, 8):
    mcs = [m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=W // groups, first_walker=i * (W // groups)) for i in range(groups)]
    def par(fn, *a):
        ts = [threading.Thread(target=getattr(mc, fn), args=a) for mc in mcs]
        [t.start() for t in ts]; [t.join() for t in ts]
    par("prepare")
    par("sweep", 1)
    t0 = time.perf_counter()
    par("sweep", 3)
    dt = (time.perf_counter() - t0) / 3
    print("groups %d: %.1f ms/sweep -> %.1f walker-sweeps/s" % (groups, dt * 1e3, W / dt), flush=True)
    for mc in mcs: mc.close()
