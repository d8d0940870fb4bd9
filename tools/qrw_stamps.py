"""In-kernel timeline of qr_rows_kernel (one workgroup of unit 0): build with `make -C montecarlo.jl_amd/csrc stamps`
(STAMP_BLOCK=<blockIdx>: 0 = part 0, owner of rows 0..63; 8 = part 1, ...), run with
DQMC_QR_ROWS=1 DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_stamps.so python tools/qrw_stamps.py"""
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
t = buf.cpu().numpy().astype(np.int64).reshape(8, 192, 8)[:4, :128, :7]
names = ["candidates+barrier(+recompute)", "select+dot+pick+npart", "publish", "collect", "scalars", "output+update"]
print("cycles (100 MHz counter: x10 ns) first to last: %d (%.1f per step)" % (t[:, -1, 6].max() - t[:, 0, 0].min(), (t[:, -1, 6].max() - t[:, 0, 0].min()) / 128))
for reg in range(4):
    sl = slice(32 * reg, 32 * reg + 32)
    print("steps %d..%d: step length %.1f" % (32 * reg, 32 * reg + 31, np.diff(t[0, :, 0])[32 * reg:32 * reg + 31].mean()))
    for w in (0, 3):
        d = np.diff(t[w, sl, :], axis=1).mean(axis=0)
        print("   wave %d: " % w + ", ".join("%s %.1f" % (n, v) for n, v in zip(names, d)))
