import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
mc.wrap_greens(5, 1)
print("ok")
