"""Determinism + health over a long run: two identical 200-sweep runs must end with bit-identical HS fields."""
import sys, os, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
m = g.load_package()
def run():
    mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32, seed=4242)
    mc.prepare()
    t0 = time.perf_counter(); mc.sweep(200); dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for w in range(32):
        h.update(mc.conf(w).tobytes())
    acc = np.mean([mc.analysis(w).acc_rate for w in range(32)])
    g5 = mc.greens_eff(5)[0].copy()
    fb, de = mc.qr_fallbacks(), mc.device_errors()
    mc.close()
    return h.hexdigest(), dt, acc, g5, fb, de
a = run(); b = run()
print("run 1: %.1f s (%.1f walker-sweeps/s), acceptance %.4f, qr fallbacks %d, device errors %d, sha %s" % (a[1], 6400 / a[1], a[2], a[4], a[5], a[0][:16]))
print("run 2: %.1f s (%.1f walker-sweeps/s), acceptance %.4f, qr fallbacks %d, device errors %d, sha %s" % (b[1], 6400 / b[1], b[2], b[4], b[5], b[0][:16]))
print("bit-identical HS fields:", a[0] == b[0], " max |dG| =", np.abs(a[3] - b[3]).max())
