import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
m = g.load_package()
rng = np.random.default_rng(0)
A = rng.standard_normal((32, 256, 256))
for it in range(2):
    out = m.udt_AVX_pivot(A)
print("ok")
