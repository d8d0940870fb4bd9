"""Cycle stamps inside the steps of one part's panel (qrb.hip, diagnostic build):
   make -C montecarlo.jl_amd/csrc stamps XFLAGS=-DQRB_FINE=<part>;  DQMC_HIP_LIB=...libdqmc_hip_stamps.so python tools/qrb_fine.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda:0")
fine = torch.zeros(4 * 32 * 8, dtype=torch.int64, device="cuda:0")
for f, b in (("dqmc_debug_qrb_stamps", buf), ("dqmc_debug_qrb_fine", fine)):
    getattr(L, f).argtypes = [C.c_void_p]
    assert getattr(L, f)(C.c_void_p(b.data_ptr())) == 0
X = np.random.default_rng(0).standard_normal((32, 256, 256))
for rep in range(3):
    gpu.udt_AVX_pivot(X, True)
torch.cuda.synchronize()
t = fine.cpu().numpy().astype(np.int64).reshape(4, 32, 8)
names = ["top->norm", "norm->scalars", "scalars->LDS", "barrier wait", "LDS+dot+sum8", "update", "raw col + helper"]
np.set_printoptions(linewidth=200)
for j in (1, 2, 3, 9, 10, 17, 18, 25, 26, 30):
    ow = j >> 3
    print("step %2d (owner wave %d): length %5d cycles" % (j, ow, t[ow, j + 1 if j < 31 else j, 0] - t[ow, j, 0] if j < 31 else 0))
    for w in range(4):
        d = np.diff(t[w, j])
        print("    wave %d%s: " % (w, "*" if w == ow else " ") + ", ".join("%s %d" % (n, v) for n, v in zip(names, d)) +
              ("   -> next top %d" % (t[w, j + 1, 0] - t[w, j, 7]) if j < 31 else ""))
