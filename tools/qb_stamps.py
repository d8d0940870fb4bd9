"""In-kernel timeline of qr_tail_kernel (unit 0): build with `make -C montecarlo.jl_amd/csrc stamps`, run with
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_stamps.so python tools/qb_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(8 * 192 * 8, dtype=torch.int64, device="cuda:0")
L.dqmc_debug_qb_stamps.argtypes = [C.c_void_p]
assert L.dqmc_debug_qb_stamps(C.c_void_p(buf.data_ptr())) == 0
rng = np.random.default_rng(0)
X = rng.standard_normal((32, 256, 256))
for rep in range(2):
    gpu.udt_AVX_pivot(X, True)
torch.cuda.synchronize()
j0 = int(os.environ.get("DQMC_QR_TAIL", "128"))
nsteps, nw = 256 - j0, (8 if j0 == 64 else 4)
t = buf.cpu().numpy().astype(np.int64).reshape(8, 192, 8)[:nw, :nsteps]
names = ["start->wave candidate", "candidate->extracted+published", "wait at barrier", "8-candidate select", "LDS column + scalars",
         "u + dots", "sum8 + update + norms"]
print("total cycles, first stamp to last: %d (%.1f per step)" % (t[:, -1, 7].max() - t[:, 0, 0].min(), (t[:, -1, 7].max() - t[:, 0, 0].min()) / nsteps))
for reg in range(nsteps // 32):
    sl = slice(32 * reg, 32 * reg + 32)
    print("region %d (steps %d..%d): step length %.0f" % (reg, 32 * reg, 32 * reg + 31, np.diff(t[0, :, 0])[32 * reg:32 * reg + 31].mean()))
    for w in (0, nw - 1):
        d = np.diff(t[w, sl, :], axis=1).mean(axis=0)
        nxt = (t[w, 1:, 0] - t[w, :-1, 7])[32 * reg:32 * reg + 31].mean()
        print("   wave %d: " % w + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, d)) + ", to next step %.0f" % nxt)
