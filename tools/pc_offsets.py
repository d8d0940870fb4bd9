"""Are the 4.5 - 6.5 sigma offsets of some pairing-correlation elements against the reference's golden (tests/golden_stats.py,
rule 1) a property of OUR pairing kernel or of the golden's std_error?  A long CPU-oracle run (python formulas of
oracle/ref_test_oracle.py, independent of the device kernel) is compared with the golden and with the device mean
(gpurun_out/pc_dev_mean.npy, written by `python tools/pc_offsets.py --device` on the GPU box: 32 walkers x 192 measurements,
the sample of tests/test_gpu_measurements.py).  Usage: python tools/pc_offsets.py [chains] [measurements per chain]"""
import os, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))


def chain(args):
    seed, n = args
    from oracle import oracle as O, ref_test_oracle as R
    mc = O.OracleDQMC(4, "attractive", beta=1.0)
    mc.set_conf(O.random_conf(seed, 16, mc.slices)); mc.seed(seed)
    mc.prepare(); mc.sweeps(50)
    out = []
    for i in range(n):
        mc.update_until_measure()
        out.append(R.pairing_correlation(mc.greens(), 4, True, 5))
    return np.array(out)


def device_mean():
    import __graft_entry__ as g
    import test_gpu_measurements as T
    s = T._device_blocks(g.load_package(), "attractive", 4, 5, 32, 24, 8, 30, 2024)
    np.save(os.path.join(R_, "gpurun_out", "pc_dev_mean.npy"), s["PC"].mean(0))


if __name__ == "__main__":
    import golden_stats as gs
    if "--device" in sys.argv:
        device_mean()
        sys.exit(0)
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    with ProcessPoolExecutor(max_workers=min(nch, 6)) as ex:
        ser = list(ex.map(chain, [(1000 + 17 * i, n) for i in range(nch)]))
    means = np.array([s.mean(0) for s in ser])
    ours = means.mean(0)
    oe = means.std(0, ddof=1) / np.sqrt(nch)          # independent chains: error of the mean of chain means
    gd = gs.load("integration_attractive_4x4.json")
    m, se = gs.golden_arrays(gd["all"]["PC"], (16, 5, 5))
    z_gold = (ours - m) / np.sqrt(se ** 2 + oe ** 2)
    print("oracle: %d chains x %d measurements; max |z| against the golden %.2f, mean z^2 %.2f, elements with |z| > 4.5: %d"
          % (nch, n, np.abs(z_gold).max(), np.mean(z_gold ** 2), int((np.abs(z_gold) > 4.5).sum())))
    dev_path = os.path.join(R_, "gpurun_out", "pc_dev_mean.npy")
    dev = np.load(dev_path) if os.path.exists(dev_path) else None
    idx = np.argsort(np.abs(z_gold).ravel())[::-1][:10]
    for i in idx:
        d, k1, k2 = np.unravel_index(i, z_gold.shape)
        line = "elem (%2d,%d,%d): golden %.6f +- %.6f | oracle %.6f +- %.6f (z %.2f)" % (d, k1, k2, m[d, k1, k2], se[d, k1, k2],
                                                                                        ours[d, k1, k2], oe[d, k1, k2], z_gold[d, k1, k2])
        if dev is not None:
            line += " | device %.6f (oracle - device = %.1f oracle sigma)" % (dev[d, k1, k2], (ours[d, k1, k2] - dev[d, k1, k2]) / oe[d, k1, k2])
        print(line)
    if dev is not None:
        zd = (ours - dev) / np.maximum(oe, 1e-300)
        print("oracle vs device mean (device error not included): max |z| %.2f, mean z^2 %.2f" % (np.abs(zd).max(), np.mean(zd ** 2)))
