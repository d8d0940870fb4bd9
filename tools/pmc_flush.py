"""Small fixed workload for PMC passes in the throughput regime: 512 units of 256 x 256 (256 repulsive walkers, beta = 1 so
that prepare() is short), 3 sweep_spatial calls = 12 elimination + 12 flush launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelRepulsive(16, 2), beta=1.0, n_walkers=256)
mc.prepare()
for _ in range(3):
    mc.sweep_spatial()
mc.close()
