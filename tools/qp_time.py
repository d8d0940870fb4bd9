"""us per batch of 256 pivoted QRs of 576 x 576 (config 5 shape; DQMC_QR_NOPANEL=1: the streaming kernel; DQMC_QP_THR:
the panel kernel's recompute threshold)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(24, 2), beta=1.0, delta_tau=0.05, n_walkers=256, seed=3)
mc.prepare()
mc.timing_enable(True)
mc.prepare()
t = mc.timing()
print(os.environ.get("DQMC_QR_NOPANEL"), os.environ.get("DQMC_QP_THR"), {k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in t.items() if v[1]})
