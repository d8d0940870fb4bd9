"""Calls udt_AVX_pivot on 32 units of n = 256 a few times (for rocprofv3 --kernel-trace --stats timing of qrb_udt_kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
gpu = g.load_package()
X = np.random.default_rng(0).standard_normal((32, 256, 256))
for rep in range(int(os.environ.get("REPS", "6"))):
    try:
        gpu.udt_AVX_pivot(X, True)
    except Exception as e:
        print("(", str(e)[:60], ")")
