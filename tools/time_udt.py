import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
mc.timing_enable(True)
mc.sweep(1)
t = mc.timing()
print(os.environ.get("DQMC_QR_STREAM"), {k: (round(v[0], 1), v[1], round(v[0]/max(v[1],1)*1e3,1)) for k, v in t.items()})
