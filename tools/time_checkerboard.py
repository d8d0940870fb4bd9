"""us per launch: sparse checkerboard product (cb.hip) against the dense GEMM with the multiplied-out constants,
config 3 shape (16x16, 32 walkers); and the whole sweep in both modes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
model = m.HubbardModelAttractive(16, 2)
for mode in ("dense", "sparse"):
    mc = m.DQMC(model, beta=8.0, delta_tau=0.1, safe_mult=10, n_walkers=32, seed=123, checkerboard=mode)
    mc.prepare(); mc.sweep(1)
    mc.timing_enable(True)
    t0 = time.perf_counter(); mc.sweep(2); dt = (time.perf_counter() - t0) / 2
    tim = mc.timing(); mc.timing_enable(False)
    print("checkerboard=%s: %.1f ms/sweep (with events); gemm family %.2f ms/sweep, %d launches, %.1f us each"
          % (mode, dt * 1e3, tim["gemm"][0] / 2, tim["gemm"][1] // 2, tim["gemm"][0] * 1e3 / tim["gemm"][1]), flush=True)
    mc.close()
