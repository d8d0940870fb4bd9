"""In-kernel timeline of trsm_rl_kernel (workgroup 0): build with `make -C montecarlo.jl_amd/csrc stamps`, run with
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_stamps.so python tools/tr_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(4 * 64, dtype=torch.int64, device="cuda:0")
L.dqmc_debug_tr_stamps.argtypes = [C.c_void_p]
assert L.dqmc_debug_tr_stamps(C.c_void_p(buf.data_ptr())) == 0
rng = np.random.default_rng(0)
n = 256
A = rng.standard_normal((32, n, n))
T = np.triu(rng.standard_normal((32, n, n))) / 16 + np.eye(n)
piv = np.tile(np.arange(1, n + 1), (32, 1))
for rep in range(3):
    gpu.rdivp(A, T, piv)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.int64).reshape(4, 64)
    s = t[0]
    print("prologue, wave 0: X tile loads issued %d, arrived %d, 3 panels requested %d, arrived %d, first panel deposited %d, barrier %d" % (
        s[40] - s[0], s[41] - s[40], s[42] - s[41], s[43] - s[42], s[44] - s[43], s[1] - s[44]))
    for w in (0, 3):
        s = t[w]
        print("wave %d: entry -> first panel in LDS %d; per step (owner + deposit + request, barrier wait, updates): %s; total %d cycles" % (
            w, s[1] - s[0], [(int(s[2 + 2 * J] - (s[1] if J == 0 else s[3 + 2 * (J - 1)])), int(s[3 + 2 * J] - s[2 + 2 * J])) for J in range(16)], s[34] - s[0]))
