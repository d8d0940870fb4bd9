"""Small fixed workload for PMC passes: 20 wrap_greens calls = 40 full batched GEMMs
(2*256^3 flops x 32 units each) and 2 sweep_spatial calls = 8 flush GEMMs after prepare()."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
for _ in range(20):
    mc.wrap_greens(5, 1)
mc.sweep_spatial(); mc.sweep_spatial()
mc.close()
