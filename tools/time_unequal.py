import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
m = g.load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = m.HubbardModelAttractive(16, 2)
mc = m.DQMC(model, beta=8.0, n_walkers=W, seed=5)
mc.set_local_targets(m.EachLocalQuadByDistance(model.l))
mc.prepare(); mc.update_until_measure()
M, s = mc.p.slices, mc.p.safe_mult
def T(fn, reps=1):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
print("sweep                         %8.1f ms" % T(lambda: mc.sweep(1)), flush=True)
mc.update_until_measure()
print("ut_build_stack (fresh conf)   %8.1f ms" % T(lambda: (mc.sweep(0), mc.ut_build_stack())))
lib = m._lib.lib()
def one(k, l): m.dqmc.DQMC._c(mc, lib.dqmc_ut_greens(mc._h, k, l, 0))
print("greens(mc, 40, 0) (stack built)%7.1f ms" % T(lambda: one(40, 0), 3))
print("greens(mc, 0, 40)              %7.1f ms" % T(lambda: one(0, 40), 3))
for rc in (s, 4 * s):
    print("susceptibilities recalc=%2d    %8.1f ms" % (rc, T(lambda: mc.accumulate_susceptibilities(rc))), flush=True)
g1 = mc.greens_kl(17, 0, 0)[0]; g2 = mc.greens_kl(17, M, 0)[0]
print("max |G(17,0) + G(17,beta)| = %.2e" % np.abs(g1 + g2).max())
it4 = list(mc.combined_greens_iterator(4 * s, walker=0)); 
ex = [mc.greens_kl(l, 0, 0)[0] for l in (5, 39, 79)]
print("CombinedGreensIterator(4 safe_mult) error vs greens(l,0):", ["%.1e" % np.abs(it4[l - 1][1][0] - e).max() for l, e in zip((5, 39, 79), ex)])
