"""config 3 (32 walkers) as 1 x 32, 2 x 16, 4 x 8 walker groups on separate handles / streams (one host thread per
handle), in phase or offset by a number of updates; and each group size alone with its per-family device times"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
FAM = ["gemm", "qr", "trsm", "sweep", "misc", "flush"]
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for W in (32, 16, 8):
    mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=W)
    mc.prepare(); mc.sweep(1)
    t0 = time.perf_counter(); mc.sweep(NS); dt = (time.perf_counter() - t0) / NS
    mc.timing_enable(True); mc.sweep(NS); tm = mc.timing(); mc.timing_enable(False)
    print("alone W=%2d: %.1f ms/sweep | " % (W, dt * 1e3) + "  ".join("%s %.1f (%d)" % (f, v[0] / NS, v[1] // NS) for f, v in tm.items()), flush=True)
    mc.close()
TOT = 32
for groups, off in ((1, 0), (2, 0), (2, 40), (2, 80), (4, 0), (4, 40)):
    mcs = [m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=TOT // groups, first_walker=i * (TOT // groups)) for i in range(groups)]
    def par(fn, *a):
        ts = [threading.Thread(target=getattr(mc, fn), args=a) for mc in mcs]
        [t.start() for t in ts]; [t.join() for t in ts]
    par("prepare")
    par("sweep", 1)
    for i, mc in enumerate(mcs):
        for _ in range(off * i):
            mc.update()
    t0 = time.perf_counter()
    par("sweep", NS)
    dt = (time.perf_counter() - t0) / NS
    print("groups %d offset %d: %.1f ms/sweep -> %.1f walker-sweeps/s" % (groups, off, dt * 1e3, TOT / dt), flush=True)
    for mc in mcs: mc.close()
