"""Workload for LDS/VALU counter passes: prepare + 2 sweep_spatial + 2 UDTs at config 3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare()
mc.sweep_spatial(); mc.sweep_spatial()
mc.close()
