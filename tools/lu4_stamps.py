"""In-kernel timeline of the elimination kernel (walker 0): build with `make -C montecarlo.jl_amd/csrc stamps`, run with
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_stamps.so python tools/lu4_stamps.py"""
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
model = gpu.HubbardModelAttractive(16, 2)
mc = gpu.DQMC(model, beta=8.0, delta_tau=0.1, safe_mult=10, n_walkers=32, seed=123)
mc.prepare()
mc.sweep(1)
for rep in range(3):
    mc.update()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.int64)
    t0 = t[360]
    rel = lambda x: (x - t0)
    piv = t[:64]
    print("kernel start->end: %d cycles" % (t[361] - t0))
    for J in range(4):
        b = 320 + 8 * J
        print(" wave %d: prologue done %d, start %d, tiles loaded %d, block ends %s, done %d" % (J, rel(t[b + 7]) if t[b + 7] else 0, rel(t[b]), rel(t[b + 1]), [int(rel(x)) if x else 0 for x in t[b + 2:b + 6]], rel(t[b + 6])))
    if t[400]:
        for J in range(4):
            b = 400 + 8 * J
            print(" wave %d prologue: entered %d, loads issued %d, loads arrived %d, solves done %d, barrier passed %d" % ((J,) + tuple(int(rel(t[b + k])) for k in range(5))))
    print(" pivot step stamps (rel):", [int(rel(x)) for x in piv[::4]])
    d = np.diff(piv)
    print(" pivot step deltas: block0 %s" % d[:15].tolist())
    print("                    block1 %s" % d[16:31].tolist())
    print("   block transitions:", int(piv[16] - piv[15]), int(piv[32] - piv[31]), int(piv[48] - piv[47]))
    for J in (1, 2, 3):  # hand-over to wave J: last decision of block J-1 -> its payload seen by J -> J's block end -> J decides
        sl = 16 * J - 1
        hJ = t[64 * (1 + J):64 * (2 + J)]
        print(" hand-over %d->%d: payload of site %d seen +%d, panel/strips done +%d, first decision +%d (cycles after that site's decision stamp)" % (
            J - 1, J, sl, hJ[sl] - piv[sl], t[320 + 8 * J + 2 + (J - 1)] - piv[sl], piv[sl + 1] - piv[sl]))
    for J in (1, 2, 3):
        h = t[64 * (1 + J):64 * (2 + J)]
        lag = [int(h[s] - piv[s]) for s in range(0, 16 * J, 3)]
        print(" helper %d lag behind the pivot's decision stamp:" % J, lag)
        print(" helper %d own site-to-site deltas, block 0:" % J, np.diff(h[:16]).tolist())
mc.close()
