"""two half-batches (2 x 16 walkers, two handles / streams) run out of phase: the second one is advanced by `off`
updates, so that its latency-bound kernels (QR, elimination) meet the other's throughput-bound ones (GEMM, flush)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
W = 32
for groups, off in ((1, 0), (2, 0), (2, 3), (2, 5), (2, 7), (4, 3)):
    mcs = [m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=W // groups, first_walker=i * (W // groups)) for i in range(groups)]
    def par(fn, *a):
        ts = [threading.Thread(target=getattr(mc, fn), args=a) for mc in mcs]
        [t.start() for t in ts]; [t.join() for t in ts]
    par("prepare")
    par("sweep", 1)
    for i, mc in enumerate(mcs):
        for _ in range(off * i):
            mc.update()
    t0 = time.perf_counter()
    par("sweep", 3)
    dt = (time.perf_counter() - t0) / 3
    print("groups %d offset %d: %.1f ms/sweep -> %.1f walker-sweeps/s" % (groups, off, dt * 1e3, W / dt), flush=True)
    for mc in mcs: mc.close()
