import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
gpu = g.load_package()
L=None
for k, m in list(sys.modules.items()):
    if k.endswith("_lib") and hasattr(m, "lib"): L = m.lib()
from oracle import oracle as O
n=256
rng=np.random.default_rng(7)
X=rng.standard_normal((1,n,n))
U,D,T,piv=gpu.udt_AVX_pivot(X, False)
dump=np.zeros(256*32)
L.dqmc_debug_qrb_dump.argtypes=[ctypes.c_void_p]
L.dqmc_debug_qrb_dump(dump.ctypes.data)
O.lib().orc_set_udt_presort(1)
Uo,Do,To,po=O.udt_pivot(X[0], False)
O.lib().orc_set_udt_presort(0)
Ro=np.triu(To)*Do[:,None]
xs=dump[:1024].reshape(256,4); di=dump[1024:2048].reshape(256,4)
np.set_printoptions(precision=4, linewidth=220)
bad=0
for tid in range(256):
    pc, rg = tid>>3, tid&7
    for k in range(4):
        row=16*(k>>1)+2*rg+(k&1)
        e=abs(xs[tid,k]-Ro[row,32+pc]); ed=abs(di[tid,k]-1/Do[row])*Do[row]
        if e>1e-9 or ed>1e-9:
            bad+=1
            if bad<12: print("tid",tid,"pc",pc,"rg",rg,"k",k,"row",row,"x",xs[tid,k],"exp",Ro[row,32+pc],"dinv",di[tid,k],"exp",1/Do[row])
print("bad x/dinv entries", bad)
E=np.abs(np.triu(T[0])-np.triu(To)); print("T bad", (E>1e-9).sum())
