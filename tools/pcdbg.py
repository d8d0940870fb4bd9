import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import golden_stats as gs
import __graft_entry__ as g
gpu = g.load_package()
import test_gpu_measurements as T
s = T._device_blocks(gpu, "attractive", 4, 5, 32, 24, 8, 30, 2024)
gd = gs.load("integration_attractive_4x4.json")
m, se = gs.golden_arrays(gd["all"]["PC"], (16, 5, 5))
ours = s["PC"].mean(0); oe = T._block_error(s["PC"])
z = np.abs(ours - m) / np.sqrt(se**2 + oe**2)
idx = np.argsort(z.ravel())[::-1][:8]
for i in idx:
    d, k1, k2 = np.unravel_index(i, z.shape)
    print("elem", (d, k1, k2), "ours %.6f +- %.6f  gold %.6f +- %.6f  z %.2f" % (ours[d,k1,k2], oe[d,k1,k2], m[d,k1,k2], se[d,k1,k2], z[d,k1,k2]))
print("mean z^2", np.mean(z**2))
np.save(os.path.join(R, "gpurun_out", "pc_dev_mean.npy"), ours)
