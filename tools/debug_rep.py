import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package(); O = g.load_oracle()
for kind, L in (("repulsive", 8), ("repulsive", 12), ("attractive", 12), ("repulsive", 16)):
    model = (m.HubbardModelRepulsive if kind == "repulsive" else m.HubbardModelAttractive)(L, 2)
    mc = m.DQMC(model, beta=1.0, n_walkers=1, seed=123)
    o = O.OracleDQMC(L, kind, beta=1.0)
    o.set_conf(mc.conf(0)); o.seed(mc.seeds[0])
    mc.prepare(); o.prepare()
    bad = None
    for u in range(6):
        mc.propagate(); o.propagate()
        mc.sweep_spatial(); o.sweep_spatial()
        c1, c2 = mc.conf(0), o.conf()
        if (c1 != c2).any():
            sl = mc.current_slice - 1
            bad = (u, np.nonzero(c1[:, sl] != c2[:, sl])[0][:5])
            break
    print(kind, L, "first divergence:", bad)
    mc.close()
