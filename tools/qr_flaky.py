"""repeat the small standalone UDT / Green's-function primitives and count mismatches against the oracle"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
gpu = g.load_package(); O = g.load_oracle()
def relerr(a, b): return np.abs(a - b).max() / np.abs(b).max()
for n in (16, 36, 64):
    rng = np.random.default_rng(n + 2)
    batch = 2
    args = []
    for _ in range(batch):
        Ul, _ = np.linalg.qr(rng.standard_normal((n, n))); Ur, _ = np.linalg.qr(rng.standard_normal((n, n)))
        Dl = np.sort(np.exp(rng.uniform(-3, 3, n)))[::-1]; Dr = np.sort(np.exp(rng.uniform(-3, 3, n)))[::-1]
        Tl = np.eye(n) + 0.1 * rng.standard_normal((n, n)); Tr = np.eye(n) + 0.1 * rng.standard_normal((n, n))
        args.append((Ul, Dl, Tl, Ur, Dr, Tr))
    stack = lambda k: np.stack([a[k] for a in args])
    ref = [O.calculate_greens(*a) for a in args]
    bad = 0; worst = 0.0
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for it in range(reps):
        G = gpu.calculate_greens_AVX(stack(0), stack(1), stack(2), stack(3), stack(4), stack(5))
        e = max(relerr(G[i], ref[i]) for i in range(batch))
        worst = max(worst, e)
        if e > 1e-10:
            bad += 1
    print("n=%d: %d / %d calls wrong, worst relerr %.3g" % (n, bad, reps, worst), flush=True)
