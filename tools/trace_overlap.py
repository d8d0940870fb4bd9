"""one sweep of config 3 under `rocprofv3 --kernel-trace`; analyse with tools/trace_overlap_parse.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
m = g.load_package()
mc = m.DQMC(m.HubbardModelAttractive(16, 2), beta=8.0, n_walkers=32)
mc.prepare(); mc.sweep(1)
mc.close()
