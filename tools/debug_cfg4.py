import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package(); O = g.load_oracle()
model = m.HubbardModelRepulsive(16, 2)
mc = m.DQMC(model, beta=8.0, n_walkers=1, seed=123)
o = O.OracleDQMC(16, "repulsive", beta=8.0)
o.set_conf(mc.conf(0)); o.seed(mc.seeds[0])
mc.prepare(); o.prepare()
for u in range(40):
    mc.propagate(); o.propagate()
    e = [np.abs(a-b).max()/np.abs(b).max() for a, b in zip(mc.greens_eff(0), o.greens_eff())]
    mc.sweep_spatial(); o.sweep_spatial()
    c1, c2 = mc.conf(0), o.conf()
    nd = int((c1 != c2).sum())
    e2 = [np.abs(a-b).max()/np.abs(b).max() for a, b in zip(mc.greens_eff(0), o.greens_eff())]
    a, st = mc.analysis(0), o.stats()
    print(u, mc.current_slice, "prop err", ["%.1e" % x for x in e], "after sweep", ["%.1e" % x for x in e2], "conf diff", nd, (a.acc_local, st.acc_local), (mc.uniforms_used(0), o.uniforms_used()))
    if nd:
        sl = mc.current_slice - 1
        print("sites differing:", np.nonzero(c1[:, sl] != c2[:, sl])[0])
        break
