"""time of the batched UDT (QR + finish + compact-WY) at config 3's shape: 32 matrices of 256 x 256 (A/B tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
mc.sweep(1)
mc.timing_enable(True)
mc.sweep(2)
t = mc.timing()
print({k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in t.items() if k in ("qr", "trsm")})
mc.close()
