"""In-kernel timeline of slab_chain_kernel (one workgroup): `make -C montecarlo.jl_amd/csrc slab_stamps`, then
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_slabstamps.so python tools/slab_stamps.py
Per wave: shader-clock cycles and 100 MHz ticks at kernel start, after the X_0 staging, after each k-loop and each write-back."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(256, dtype=torch.int64, device="cuda:0")
L.dqmc_debug_slab_stamps.argtypes = [C.c_void_p]
assert L.dqmc_debug_slab_stamps(C.c_void_p(buf.data_ptr())) == 0
mc = gpu.DQMC(gpu.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32, seed=3)
mc.prepare()
for rep in range(3):
    mc.wrap_greens(5, 1)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.int64).reshape(4, 32, 2)
    for w in range(4):
        cyc, rt = t[w, :6, 0], t[w, :6, 1]
        dc, dr = cyc - cyc[0], (rt - rt[0]) * 10
        print("wrap rep %d wave %d: cycles %s | ns %s | clock %.2f GHz | k-loops %s cycles" % (
            rep, w, dc.tolist(), dr.tolist(), dc[5] / max(dr[5], 1), [int(cyc[2] - cyc[1]), int(cyc[4] - cyc[3])]), flush=True)
        if t[w, 20, 0]:
            print("    prologue: requests issued %d, slab arrived %d, slab in LDS %d, barrier passed %d (cycles after start)" % (
                t[w, 20, 0] - cyc[0], t[w, 21, 0] - cyc[0], t[w, 22, 0] - cyc[0], cyc[1] - cyc[0]), flush=True)
mc.close()
