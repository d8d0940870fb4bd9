import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
for wg in (1, 2, 3, 4, 6, 8):
    os.environ["DQMC_PROBE_WG_PER_CU"] = str(wg)
    print(wg, "workgroups of 4 waves per CU: %.1f TF/s" % m.mfma_f64_peak(20000), flush=True)
