import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=int(os.environ.get("W", "32")))
mc.prepare()
# wrap_greens = 2 full GEMMs per call, shared-constant operand on one side
mc.wrap_greens(5, 1)
mc.timing_enable(True)
for _ in range(50): mc.wrap_greens(5, 1)
t = mc.timing()["gemm"]
us = t[0] / t[1] * 1e3
print("full gemm: %.1f us  -> %.1f TFLOP/s" % (us, 2 * 256**3 * mc.n_walkers / us / 1e6))
