import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
gpu = g.load_package()
import importlib
L = sys.modules[gpu.__name__ + "._lib"].lib() if (gpu.__name__ + "._lib") in sys.modules else None
if L is None:
    for k, m in sys.modules.items():
        if k.endswith("_lib") and hasattr(m, "lib"): L = m.lib()
from oracle import oracle as O
n=256
rng=np.random.default_rng(7)
X=rng.standard_normal((1,n,n))
U,D,T,piv=gpu.udt_AVX_pivot(X, False)
dump=np.zeros(256*32)
L.dqmc_debug_qrb_dump.argtypes=[ctypes.c_void_p]
print("rc", L.dqmc_debug_qrb_dump(dump.ctypes.data))
Cd=dump.reshape((256,32),order='F')
Xs=X[0][:, piv[0]-1]
# expected: apply the 32 reflectors of panel 0
import scipy.linalg as sl
Q,R=np.linalg.qr(Xs[:, :32], mode='complete')
exp=Q.T@Xs[:, 32:64]
# sign convention: rows of Q^T may differ in sign from Householder (R diag sign) -> compare abs for rows<32, and for rows>=32 compare the projected subspace norm
np.set_printoptions(precision=4, linewidth=220)
err_rows=np.abs(np.abs(Cd[:32])-np.abs(exp[:32])).max(axis=1)
print("rows<32 |abs| err per row", err_rows)
# trailing block: compare Gram matrices (invariant to orthogonal transformation of rows>=32)
G1=Cd[32:].T@Cd[32:]; G2=exp[32:].T@exp[32:]
print("trailing gram err", np.abs(G1-G2).max())
