import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
gpu = g.load_package()
from oracle import oracle as O
n=256
rng=np.random.default_rng(7)
X=rng.standard_normal((1,n,n))
for ap in (True, False):
    U,D,T,piv=gpu.udt_AVX_pivot(X, ap)
    O.lib().orc_set_udt_presort(1)
    Uo,Do,To,po=O.udt_pivot(X[0], ap)
    O.lib().orc_set_udt_presort(0)
    print("apply",ap,"piv equal", np.array_equal(po,piv[0]), "D err", np.abs(D[0]-Do).max()/Do.max(), "U err", np.abs(U[0]-Uo).max())
    Tt = T[0] if ap else np.triu(T[0]); Tto = To if ap else np.triu(To)
    E=np.abs(Tt-Tto)
    if not ap:
        bad=np.argwhere(E>1e-9)
        print("bad count",len(bad)); print(bad[:20]); 
        rows=np.unique(bad[:,0]); cols=np.unique(bad[:,1]); print("rows",rows[:40]); print("cols", cols[:40])
        if len(bad): 
            i,j=bad[0]; print(Tt[i,j], Tto[i,j], Tt[i,j]/Tto[i,j])
    else:
        # unpermute columns: T[:, piv[j]-1] = column j
        Eu=E[:, piv[0]-1]
        bad=np.argwhere(Eu>1e-9); print("bad count",len(bad)); print(bad[:20])
        rows=np.unique(bad[:,0]); cols=np.unique(bad[:,1]); print("rows",rows[:40]); print("cols", cols[:40])
    Tn = (Tt[:, piv[0]-1] if ap else Tt); Tn2 = (Tto[:, piv[0]-1] if ap else Tto)
    np.set_printoptions(precision=4, linewidth=220)
    print("ours*D rows 10..17, cols 32..43"); print(Tn[10:18, 32:44]*Do[10:18,None])
    print("oracle*D"); print(Tn2[10:18, 32:44]*Do[10:18,None])
    print("X cols (presorted) rows 10..17"); print(X[0][:, po-1][10:18, 32:44])
