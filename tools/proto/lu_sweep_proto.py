"""numpy prototype of the round-2 sweep algebra (tools only, not shipped):
   A: conditional trailing elimination on S = G[c,c] (+ PT_J, Q_J by rank-1 recurrences)
   B: block-triangular solves + flush, compared with literal sequential rank-1 updates."""
import numpy as np
rng = np.random.default_rng(1)
n, KD, site0 = 256, 64, 64
G0 = 0.5 * np.eye(n) + 0.1 * rng.standard_normal((n, n))
gam = np.where(rng.random(KD) < 0.5, 0.88, -0.47)
accept = rng.random(KD) < 0.8

# literal sequential
G = G0.copy()
xs_ref = np.zeros(KD)
for s in range(KD):
    i = site0 + s
    if not accept[s]:
        continue
    r = 1.0 + gam[s] * (1.0 - G[i, i])
    x = gam[s] / r
    xs_ref[s] = x
    IG = -G[:, i].copy(); IG[i] += 1.0
    G -= np.outer(IG * x, G[i, :])

# ---- phase A on the 64x64 block
c = slice(site0, site0 + KD)
S = G0[c, c].copy()
xs = np.zeros(KD)
PT = [np.eye(16) for _ in range(4)]
Q = [np.eye(16) for _ in range(4)]
for s in range(KD):
    if not accept[s]:
        continue
    d = S[s, s]
    x = gam[s] / (1.0 + gam[s] * (1.0 - d))
    xs[s] = x
    I0, cc = s // 16, s % 16
    v = S[s, :].copy(); v[: s + 1] = 0.0          # row s, later columns
    u = S[:, s].copy(); u[: s + 1] = 0.0          # column s, later rows
    S += np.outer(x * u, v)                       # trailing update only (rows, cols > s)
    vb = v[16 * I0: 16 * I0 + 16]; ub = u[16 * I0: 16 * I0 + 16]
    PT[I0] += np.outer(x * vb, PT[I0][cc, :])     # PT[j][i] += x v[j] PT[m][i]
    Q[I0] += np.outer(x * ub, Q[I0][cc, :])       # Q[k][j] += x ucol[k] Q[m][j]
assert np.allclose(xs, xs_ref, rtol=1e-12, atol=1e-14), np.abs(xs - xs_ref).max()
F = S  # compact: strict upper = Uu (rows of accepted sites), strict lower = L
X = np.diag(xs)
Uu = np.triu(F, 1); L = np.tril(F, -1)
for J in range(4):
    b = slice(16 * J, 16 * J + 16)
    assert np.allclose(PT[J], np.linalg.inv(np.eye(16) - X[b, b] @ Uu[b, b]).T)
    assert np.allclose(Q[J], np.linalg.inv(np.eye(16) - L[b, b] @ X[b, b]))

# ---- phase B
C0T = G0[:, c].T.copy()
C0T[np.arange(KD), site0 + np.arange(KD)] -= 1.0   # C0^T[s][t] = G0[t][site0+s] - delta
R0 = G0[c, :].copy()
blk = lambda J: slice(16 * J, 16 * J + 16)
Z = [None] * 4; XZ = [None] * 4
for J in range(4):
    acc = C0T[blk(J), :].copy()
    for K in range(J):
        acc += F[blk(K), blk(J)].T @ XZ[K]
    Z[J] = PT[J] @ acc
    XZ[J] = X[blk(J), blk(J)] @ Z[J]
TT = [None] * 4
for J in range(3, -1, -1):
    acc = np.zeros((16, n))
    for K in range(J + 1, 4):
        acc += F[blk(K), blk(J)].T @ TT[K]
    acc = XZ[J] + X[blk(J), blk(J)] @ acc
    TT[J] = Q[J].T @ acc
Tt = np.vstack(TT)                                 # T^T (64 x n)
Gnew = G0 + Tt.T @ R0
print("max |Gnew - Gseq| =", np.abs(Gnew - G).max(), " max|G| =", np.abs(G).max())
assert np.allclose(Gnew, G, rtol=1e-10, atol=1e-12)
print("OK")
