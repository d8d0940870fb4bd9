import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
gpu = g.load_package()
from oracle import oracle as O
from test_gpu_udt_blocked import graded_cases
n=256
X=graded_cases(n, np.random.default_rng(7))
for ap in (True, False):
    U,D,T,piv=gpu.udt_AVX_pivot(X, ap)
    for i in range(len(X)):
        O.lib().orc_set_udt_presort(1); Uo,Do,To,po=O.udt_pivot(X[i], ap); O.lib().orc_set_udt_presort(0)
        eq=np.array_equal(po,piv[i])
        Tt, Tto = (T[i], To) if ap else (np.triu(T[i]), np.triu(To))
        if ap: rec=(U[i]*D[i])@T[i]
        else:
            P=np.zeros((n,n)); P[np.arange(n), piv[i]-1]=1; rec=(U[i]*D[i])@np.triu(T[i])@P
        scale=np.abs(X[i]).max(axis=0)
        print("ap",ap,"case",i,"piv eq",eq,"D max-rel %.1e elem-rel %.1e"%(np.abs(D[i]-Do).max()/Do.max(), np.abs(D[i]/Do-1).max()),
              "U %.1e"%np.abs(U[i]-Uo).max(), "T %.1e"%(np.abs(Tt-Tto).max()/np.abs(Tto).max()), "orth %.1e"%np.abs(U[i].T@U[i]-np.eye(n)).max(),
              "rec %.1e"%(np.abs(rec-X[i])/scale[None,:]).max(), "D sorted", bool(np.all(np.diff(D[i])<=0)), "maxT %.1e"%np.abs(Tt).max())
