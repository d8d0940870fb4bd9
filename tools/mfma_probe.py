import sys, os, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
for it in (20000, 200000, 1000000, 1000000):
    t0 = time.perf_counter()
    tf = m.mfma_f64_peak(it)
    print("iters %d: %.1f TFLOP/s (wall %.3f s)" % (it, tf, time.perf_counter() - t0), flush=True)
