"""In-kernel timeline of one flush workgroup in the throughput regime (512 units; workgroup FL_STAMP_BID, default 1200 = third
round): build with `make -C montecarlo.jl_amd/csrc stamps`, run with
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_stamps.so python tools/fl_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(512, dtype=torch.int64, device="cuda:0")
L.dqmc_debug_lu4_stamps.argtypes = [C.c_void_p]
assert L.dqmc_debug_lu4_stamps(C.c_void_p(buf.data_ptr())) == 0
mc = gpu.DQMC(gpu.HubbardModelRepulsive(16, 2), beta=1.0, n_walkers=256, seed=3)
mc.prepare()
for rep in range(3):
    mc.sweep_spatial()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.int64)[480:492]
    print("entry -> operands in LDS %d, solves %d, barrier %d, passes (MFMA + store + next tile to LDS + barrier) %s, total %d cycles" % (
        t[1] - t[0], t[2] - t[1], t[3] - t[2], np.diff(t[3:8]).tolist(), t[7] - t[0]))
