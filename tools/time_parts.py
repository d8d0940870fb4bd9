"""A/B timing of parts of the engine on the GPU box (one tool instead of a script per question).
  python tools/time_parts.py families [walkers] [attractive|repulsive]   per-family device ms of one sweep (config 3 / 4 shape)
  python tools/time_parts.py wrap [walkers]                              us per wrap_greens launch, TFLOP/s
  python tools/time_parts.py sweep_spatial                               us per launch of the site-sweep phase
  python tools/time_parts.py checkerboard                                dense constants against the sparse bond-group kernel
  python tools/time_parts.py soak [sweeps]                               two identical long runs: bit-identical HS fields?
Kernel-selection switches are environment variables read per handle."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
m = g.load_package()
what = sys.argv[1] if len(sys.argv) > 1 else "families"
arg = sys.argv[2:]


def families():
    W = int(arg[0]) if arg else 32
    Model = m.HubbardModelRepulsive if (len(arg) > 1 and arg[1] == "repulsive") else m.HubbardModelAttractive
    mc = m.DQMC(Model(16, 2), beta=8.0, n_walkers=W, seed=1)
    mc.prepare(); mc.sweep(1)
    t0 = time.perf_counter(); mc.sweep(3); dt = (time.perf_counter() - t0) / 3
    mc.timing_enable(True); mc.sweep(2); t = mc.timing()
    print("%d walkers: %.2f ms/sweep -> %.1f walker-sweeps/s | device ms per sweep (launches, us each): %s | qr fallbacks %d" % (
        W, dt * 1e3, W / dt, {k: (round(v[0] / 2, 2), v[1] // 2, round(v[0] / max(v[1], 1) * 1e3, 1)) for k, v in t.items()},
        mc.qr_fallbacks()))
    mc.close()


def wrap():
    W = int(arg[0]) if arg else 32
    mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=W)
    mc.prepare(); mc.wrap_greens(5, 1)
    mc.timing_enable(True)
    for _ in range(50): mc.wrap_greens(5, 1)
    t = mc.timing()["gemm"]
    us = t[0] / t[1] * 1e3
    print("wrap_greens launch (2 products): %.1f us -> %.1f TFLOP/s" % (us, 4 * 256 ** 3 * W / us / 1e6))
    mc.close()


def sweep_spatial():
    mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
    mc.prepare(); mc.sweep_spatial()
    mc.timing_enable(True)
    for _ in range(20): mc.sweep_spatial()
    print({k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in mc.timing().items() if v[1]})
    mc.close()


def checkerboard():
    model = m.HubbardModelAttractive(16, 2)
    for mode in ("dense", "sparse"):
        mc = m.DQMC(model, beta=8.0, n_walkers=32, seed=123, checkerboard=mode)
        mc.prepare(); mc.sweep(1)
        mc.timing_enable(True)
        t0 = time.perf_counter(); mc.sweep(2); dt = (time.perf_counter() - t0) / 2
        tim = mc.timing()
        print("checkerboard=%s: %.1f ms/sweep (with events); gemm family %.2f ms/sweep, %d launches, %.1f us each"
              % (mode, dt * 1e3, tim["gemm"][0] / 2, tim["gemm"][1] // 2, tim["gemm"][0] * 1e3 / tim["gemm"][1]), flush=True)
        mc.close()


def soak():
    ns = int(arg[0]) if arg else 200

    def run():
        mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32, seed=4242)
        mc.prepare()
        t0 = time.perf_counter(); mc.sweep(ns); dt = time.perf_counter() - t0
        h = hashlib.sha256()
        for w in range(32):
            h.update(mc.conf(w).tobytes())
        acc = np.mean([mc.analysis(w).acc_rate for w in range(32)])
        g5, fb = mc.greens_eff(5)[0].copy(), mc.qr_fallbacks()
        mc.close()
        return h.hexdigest(), dt, acc, g5, fb
    a, b = run(), run()
    for i, r in enumerate((a, b)):
        print("run %d: %.1f s (%.1f walker-sweeps/s), acceptance %.4f, qr fallbacks %d, sha %s" % (i + 1, r[1], 32 * ns / r[1], r[2], r[4], r[0][:16]))
    print("bit-identical HS fields:", a[0] == b[0], " max |dG| =", np.abs(a[3] - b[3]).max())


{"families": families, "wrap": wrap, "sweep_spatial": sweep_spatial, "checkerboard": checkerboard, "soak": soak}[what]()
