import sys, os, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
mc.sweep_spatial()
mc.timing_enable(True)
for _ in range(20): mc.sweep_spatial()
t = mc.timing()
print(os.environ.get("DQMC_DEBUG_SWEEP"), {k: (round(v[0]/max(v[1],1)*1e3,1), v[1]) for k, v in t.items()})
