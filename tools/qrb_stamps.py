"""In-kernel timeline of qrb_udt_kernel (the eight parts of unit 0): build with `make -C montecarlo.jl_amd/csrc -f diag.mk qrb_stamps`, run with
DQMC_HIP_LIB=montecarlo.jl_amd/libdqmc_hip_qrbstamps.so python tools/qrb_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
gpu = g.load_package()
L = gpu.lib()
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda:0")
L.dqmc_debug_qrb_stamps.argtypes = [C.c_void_p]
assert L.dqmc_debug_qrb_stamps(C.c_void_p(buf.data_ptr())) == 0
rng = np.random.default_rng(0)
X = rng.standard_normal((32, 256, 256))
for rep in range(3):
    gpu.udt_AVX_pivot(X, True)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.int64).reshape(8, 64)
t0 = t[:, 0].min()
f = 100.0 / 1e3  # s_memtime counts at 100 MHz: 10 ns per tick -> print in us
names = {0: "start", 1: "norms out", 2: "norms in", 3: "ranked", 4: "gathered", 48: "own begin", 49: "converted", 50: "step 8", 51: "step 16",
         52: "step 24", 53: "step 32", 54: "T written", 60: "end"}
for p in range(8):
    names[8 + 4 * p] = "p%d fetch" % p
    names[9 + 4 * p] = "p%d seen" % p
    names[10 + 4 * p] = "p%d staged" % p
    names[11 + 4 * p] = "p%d applied" % p
print("times in us from the first start; s_memtime tick = %s" % os.environ.get("QRB_TICK_NS", "10 ns"))
tick = float(os.environ.get("QRB_TICK_NS", "10")) / 1e3
for part in range(8):
    ev = sorted((int(t[part, k]), k) for k in range(64) if t[part, k] > 0)
    print("part %d: " % part + "  ".join("%s %.2f" % (names.get(k, str(k)), (v - t0) * tick) for v, k in ev))
