"""Independent oracles restated from the reference's own TESTS. TEST INFRASTRUCTURE ONLY.

1. The LAPACK-QR Green's function oracle of test/testfunctions.jl:10-118 with
   decompose_udt of test/linalg/old_linalg.jl:16-24.  Julia's qr(A, Val(true))
   is LAPACK dgeqp3, which scipy.linalg.qr(pivoting=True) calls as well, so this
   is an implementation of G that shares no code with oracle/dqmc_oracle.c
   (different QR, different pivot rule, dense inverses).
2. Exact diagonalisation of the 2x2 Hubbard model, test/ED/ED.jl:68-120 (Hamiltonian)
   and :497-518 (equal-time Green's function), used at test/ED/ED_tests.jl:91-176.
"""
import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------
# test/linalg/old_linalg.jl:16-24
def decompose_udt(A):
    Q, R, p = sla.qr(A, pivoting=True)
    pinv = np.empty_like(p)
    pinv[p] = np.arange(len(p))
    D = np.abs(np.diag(R))
    T = (R / D[:, None])[:, pinv]
    return Q, D, T


def slice_matrix(eT2, lam, conf, slice_, sign=+1.0):
    """slice_matrices.jl:10-21 for the attractive model; slice_ is 1-based.
    `sign` = +1 gives block 0 / attractive, -1 the second (down) block of the
    repulsive model (Repulsive.jl:113-126)."""
    eV = np.exp(sign * lam * conf[:, slice_ - 1].astype(np.float64))
    return eT2 * eV[None, :]


# test/testfunctions.jl:10-42: Ul,Dl,Tl = B(stop)...B(start)
def slice_matrix_chain(B, start, stop, safe_mult):
    n = B(1).shape[0]
    U = np.eye(n); D = np.ones(n); T = np.eye(n)
    for k in range(start, stop + 1):
        U = B(k) @ U
        if k % safe_mult == 0:
            U = U * D[None, :]
            U, D, Tnew = decompose_udt(U)
            T = Tnew @ T
    U = U * D[None, :]
    U, D, Tnew = decompose_udt(U)
    T = Tnew @ T
    return U, D, T


# test/testfunctions.jl:45-77: Ur,Dr,Tr = B(start)' ... B(stop)'
def slice_matrix_chain_dagger(B, start, stop, safe_mult):
    n = B(1).shape[0]
    U = np.eye(n); D = np.ones(n); T = np.eye(n)
    for k in range(stop, start - 1, -1):
        U = B(k).T @ U
        if k % safe_mult == 0:
            U = U * D[None, :]
            U, D, Tnew = decompose_udt(U)
            T = Tnew @ T
    U = U * D[None, :]
    U, D, Tnew = decompose_udt(U)
    T = Tnew @ T
    return U, D, T


# test/testfunctions.jl:80-118: G(slice) = [1 + B(slice)...B(1) B(M)...B(slice+1)]^-1
def calculate_greens_and_logdet(B, M, slice_, safe_mult):
    n = B(1).shape[0]
    if slice_ + 1 <= M:
        Ur, Dr, Tr = slice_matrix_chain_dagger(B, slice_ + 1, M, safe_mult)
    else:
        Ur, Dr, Tr = np.eye(n), np.ones(n), np.eye(n)
    if slice_ >= 1:
        Ul, Dl, Tl = slice_matrix_chain(B, 1, slice_, safe_mult)
    else:
        Ul, Dl, Tl = np.eye(n), np.ones(n), np.eye(n)
    tmp = Tl @ Tr.T
    U, D, T = decompose_udt(Dl[:, None] * tmp * Dr[None, :])
    U = Ul @ U
    T = T @ Ur.T
    u, d, t = decompose_udt(U.T @ np.linalg.inv(T) + np.diag(D))
    T = np.linalg.inv(t @ T)
    U = (U @ u).T
    d = 1.0 / d
    return T @ np.diag(d) @ U


# --------------------------------------------------------------------------
# Exact diagonalisation, test/ED/ED.jl
def _fermion_ops(n_modes):
    """Jordan-Wigner annihilators c_0..c_{n-1} on the 2^n Fock space."""
    I2 = np.eye(2); Z = np.diag([1.0, -1.0]); a = np.array([[0.0, 1.0], [0.0, 0.0]])
    ops = []
    for m in range(n_modes):
        mats = [Z] * m + [a] + [I2] * (n_modes - m - 1)
        M = mats[0]
        for x in mats[1:]:
            M = np.kron(M, x)
        ops.append(M)
    return ops


def ed_hubbard_greens(neighs, n_sites, U, t, mu, beta, return_state=False):
    """G[(s1,i),(s2,j)] = <c_{i,s1} c^dagger_{j,s2}> for
    H = -t sum_{src, trg in neighs[:,src], sigma} c^dag_{trg} c_{src}
        + U sum_i (n_up-1/2)(n_dn-1/2) - mu sum_i n_i      (ED.jl:68-120);
    U < 0 for the attractive model.  neighs is 4 x N, 1-based.  Mode index =
    n_sites*substate + site as in calculate_Greens_matrix (ED.jl:497-518)."""
    nm = 2 * n_sites
    c = _fermion_ops(nm)
    cd = [x.T for x in c]
    H = np.zeros((2 ** nm, 2 ** nm))
    for s in range(2):
        for src in range(n_sites):
            for trg in neighs[:, src] - 1:
                H -= t * cd[n_sites * s + trg] @ c[n_sites * s + src]
    Id = np.eye(2 ** nm)
    for i in range(n_sites):
        nu = cd[i] @ c[i]
        nd = cd[n_sites + i] @ c[n_sites + i]
        H += U * (nu - 0.5 * Id) @ (nd - 0.5 * Id) - mu * (nu + nd)
    w, V = np.linalg.eigh(H)
    w = w - w.min()
    rho = (V * np.exp(-beta * w)) @ V.T
    Z = np.trace(rho)
    if return_state:
        return rho / Z, c, cd
    G = np.zeros((nm, nm))
    for a_ in range(nm):
        for b_ in range(nm):
            G[a_, b_] = np.trace(rho @ c[a_] @ cd[b_]) / Z
    return G


# --------------------------------------------------------------------------
# Equal-time measurement kernels in their generic 2N x 2N form
# (src/flavors/DQMC/measurements/measurements.jl:51-190) and the pair iterator
# (src/lattices/lattice_iterators.jl:137-190), restated independently of the product.
def full_greens(blocks):
    """greens(mc, model): the 2N x 2N matrix (attractive: blockdiag(G, G), Attractive.jl:169-172)"""
    n = blocks[0].shape[0]
    G = np.zeros((2 * n, 2 * n))
    G[:n, :n] = blocks[0]
    G[n:, n:] = blocks[1] if len(blocks) == 2 else blocks[0]
    return G


def cdc_kernel(G, N, i, j):
    d = 1.0 if i == j else 0.0
    return ((1 - G[i, i]) * (1 - G[j, j]) + (d - G[j, i]) * G[i, j]
            + (1 - G[i, i]) * (1 - G[j + N, j + N]) - G[j + N, i] * G[i, j + N]
            + (1 - G[i + N, i + N]) * (1 - G[j, j]) - G[j, i + N] * G[i + N, j]
            + (1 - G[i + N, i + N]) * (1 - G[j + N, j + N]) + (d - G[j + N, i + N]) * G[i + N, j + N])


def sdc_x_kernel(G, N, i, j):
    d = 1.0 if i == j else 0.0
    return (G[i + N, i] * G[j + N, j] - G[j + N, i] * G[i + N, j]
            + G[i + N, i] * G[j, j + N] + (d - G[j, i]) * G[i + N, j + N]
            + G[i, i + N] * G[j + N, j] + (d - G[j + N, i + N]) * G[i, j]
            + G[i, i + N] * G[j, j + N] - G[j, i + N] * G[i, j + N])


def sdc_y_kernel(G, N, i, j):
    d = 1.0 if i == j else 0.0
    return (-G[i + N, i] * G[j + N, j] + G[j + N, i] * G[i + N, j]
            + G[i + N, i] * G[j, j + N] + (d - G[j, i]) * G[i + N, j + N]
            + G[i, i + N] * G[j + N, j] + (d - G[j + N, i + N]) * G[i, j]
            - G[i, i + N] * G[j, j + N] + G[j, i + N] * G[i, j + N])


def sdc_z_kernel(G, N, i, j):
    d = 1.0 if i == j else 0.0
    return ((1 - G[i, i]) * (1 - G[j, j]) + (d - G[j, i]) * G[i, j]
            - (1 - G[i, i]) * (1 - G[j + N, j + N]) + G[j + N, i] * G[i, j + N]
            - (1 - G[i + N, i + N]) * (1 - G[j, j]) + G[j, i + N] * G[i + N, j]
            + (1 - G[i + N, i + N]) * (1 - G[j + N, j + N]) + (d - G[j + N, i + N]) * G[i + N, j + N])


def square_pair_directions(L, eps=1e-6):
    """EachSitePairByDistance(SquareLattice(L)): list of displacement vectors (sorted) and the
    0-based direction index of every (src, trg) pair, src/trg 0-based column-major sites"""
    pos = [np.array([i + 1.0, j + 1.0]) for j in range(L) for i in range(L)]
    wrap = [np.zeros(2)]
    for v in (np.array([float(L), 0.0]), np.array([0.0, float(L)])):
        wrap = [e - v for e in wrap] + wrap + [e + v for e in wrap]

    def dnorm(v):
        ln = np.hypot(v[0], v[1])
        if ln > eps:
            a = np.arccos(v[0] / ln)
            if v[1] < 0:
                a = 2 * np.pi - a
            return ln + eps * a
        return ln

    dirs, first = [], {}
    table = np.zeros((L * L, L * L), dtype=np.int64)
    for o in range(L * L):
        for t in range(L * L):
            d = pos[o] - pos[t] + wrap[0]
            for v in wrap[1:]:
                nd = pos[o] - pos[t] + v
                if dnorm(nd) + eps < dnorm(d):
                    d = nd
            key = (int(round(d[0])), int(round(d[1])))
            if key not in first:
                first[key] = len(dirs)
                dirs.append(d.copy())
            table[o, t] = first[key]
    order = sorted(range(len(dirs)), key=lambda k: dnorm(dirs[k]))
    rank = np.empty(len(dirs), dtype=np.int64)
    rank[order] = np.arange(len(dirs))
    return [dirs[k] for k in order], rank[table]


def equal_time_correlations(blocks, L, attractive):
    """what apply!(EachSitePairByDistance, ...) + finish! push for CDC, SDCx/y/z (generic.jl:325-330,
    283-286) and m{x,y,z}_kernel per site, from the block Green's functions of one configuration"""
    N = L * L
    G = full_greens(blocks)
    dirs, table = square_pair_directions(L)
    nd = len(dirs)
    out = {k: np.zeros(nd) for k in ("CDC", "SDCx", "SDCy", "SDCz")}
    for i in range(N):
        for j in range(N):
            d = table[i, j]
            if attractive:  # HubbardModelAttractive.jl:222-236 overrides
                dij = 1.0 if i == j else 0.0
                t = 2 * (dij - G[j, i]) * G[i, j]
                out["CDC"][d] += 4 * (1 - G[i, i]) * (1 - G[j, j]) + t
                out["SDCx"][d] += t; out["SDCy"][d] += t; out["SDCz"][d] += t
            else:
                out["CDC"][d] += cdc_kernel(G, N, i, j)
                out["SDCx"][d] += sdc_x_kernel(G, N, i, j)
                out["SDCy"][d] += sdc_y_kernel(G, N, i, j)
                out["SDCz"][d] += sdc_z_kernel(G, N, i, j)
    for k in out:
        out[k] /= N
    z = np.zeros(N)
    out["Mx"] = z.copy() if attractive else np.array([-G[i + N, i] - G[i, i + N] for i in range(N)])
    out["My"] = z.copy() if attractive else np.array([G[i + N, i] - G[i, i + N] for i in range(N)])
    out["Mz"] = z.copy() if attractive else np.array([G[i + N, i + N] - G[i, i] for i in range(N)])
    return out


# --------------------------------------------------------------------------
# pairing correlation: pc_kernel (measurements.jl:208-214) over EachLocalQuadByDistance{K}
# (lattice_iterators.jl:264-318, apply!/finish! generic.jl:287-290,341-349)
def pc_kernel(G, N, src1, trg1, src2, trg2):
    """<Delta_v(src1, trg1) Delta_v^dagger(src2, trg2)> by Wick's theorem, 0-based sites"""
    return G[src1, src2] * G[trg1 + N, trg2 + N] - G[src1, trg2 + N] * G[trg1 + N, src2]


def ed_pairing(rho, c, cd, N, src1, trg1, src2, trg2):
    """the same expectation value taken directly in the Fock space:
    Delta(i, j) = c_{i,up} c_{j,dn},  Delta^dagger(i, j) = c^dag_{j,dn} c^dag_{i,up}"""
    op = c[src1] @ c[N + trg1] @ cd[N + trg2] @ cd[src2]
    return np.trace(rho @ op)


def pairing_correlation(blocks, L, attractive, K):
    """output[dir12, dir1, dir2] as pushed by finish!, from the block Green's functions of one
    configuration.  The (dir, trg) lists are rebuilt here from square_pair_directions."""
    N = L * L
    G = full_greens(blocks)
    dirs, table = square_pair_directions(L)
    nd = len(dirs)
    trg = [[[t for t in range(N) if table[s, t] == k] for k in range(K)] for s in range(N)]
    out = np.zeros((nd, K, K))
    for s1 in range(N):
        for s2 in range(N):
            d12 = table[s1, s2]
            for k1 in range(K):
                for t1 in trg[s1][k1]:
                    for k2 in range(K):
                        for t2 in trg[s2][k2]:
                            if attractive:  # HubbardModelAttractive.jl:243-245
                                out[d12, k1, k2] += G[s1, s2] * G[t1, t2]
                            else:
                                out[d12, k1, k2] += pc_kernel(G, N, s1, t1, s2, t2)
    return out / N


# --------------------------------------------------------------------------
# packed-Green's-function kernels (G00, G0l, Gl0, Gll) in their generic 2N x 2N form
# (measurements.jl:76-92, 158-192, 215-219) and the susceptibility sums of
# apply!(::CombinedGreensIterator, ...) + finish!(..., delta_tau) (generic.jl:226-243, 283-290)
def cdc_kernel_packed(pg, N, i, j):
    G00, G0l, Gl0, Gll = pg
    return ((1 - Gll[i, i]) * (1 - G00[j, j]) - G0l[j, i] * Gl0[i, j]
            + (1 - Gll[i, i]) * (1 - G00[j + N, j + N]) - G0l[j + N, i] * Gl0[i, j + N]
            + (1 - Gll[i + N, i + N]) * (1 - G00[j, j]) - G0l[j, i + N] * Gl0[i + N, j]
            + (1 - Gll[i + N, i + N]) * (1 - G00[j + N, j + N]) - G0l[j + N, i + N] * Gl0[i + N, j + N])


def sdc_x_kernel_packed(pg, N, i, j):
    G00, G0l, Gl0, Gll = pg
    return (Gll[i + N, i] * G00[j + N, j] - G0l[j + N, i] * Gl0[i + N, j]
            + Gll[i + N, i] * G00[j, j + N] - G0l[j, i] * Gl0[i + N, j + N]
            + Gll[i, i + N] * G00[j + N, j] - G0l[j + N, i + N] * Gl0[i, j]
            + Gll[i, i + N] * G00[j, j + N] - G0l[j, i + N] * Gl0[i, j + N])


def sdc_y_kernel_packed(pg, N, i, j):
    G00, G0l, Gl0, Gll = pg
    return (-Gll[i + N, i] * G00[j + N, j] + G0l[j + N, i] * Gl0[i + N, j]
            + Gll[i + N, i] * G00[j, j + N] - G0l[j, i] * Gl0[i + N, j + N]
            + Gll[i, i + N] * G00[j + N, j] - G0l[j + N, i + N] * Gl0[i, j]
            - Gll[i, i + N] * G00[j, j + N] + G0l[j, i + N] * Gl0[i, j + N])


def sdc_z_kernel_packed(pg, N, i, j):
    G00, G0l, Gl0, Gll = pg
    return ((1 - Gll[i, i]) * (1 - G00[j, j]) - G0l[j, i] * Gl0[i, j]
            - (1 - Gll[i, i]) * (1 - G00[j + N, j + N]) + G0l[j + N, i] * Gl0[i, j + N]
            - (1 - Gll[i + N, i + N]) * (1 - G00[j, j]) + G0l[j, i + N] * Gl0[i + N, j]
            + (1 - Gll[i + N, i + N]) * (1 - G00[j + N, j + N]) - G0l[j + N, i + N] * Gl0[i + N, j + N])


def pc_kernel_packed(pg, N, src1, trg1, src2, trg2):
    Gl0 = pg[2]
    return Gl0[src1, src2] * Gl0[trg1 + N, trg2 + N] - Gl0[src1, trg2 + N] * Gl0[trg1 + N, src2]


def susceptibilities(g00_blocks, steps, L, attractive, K, delta_tau):
    """steps: iterable over l of (G0l, Gl0, Gll), each a list of per-block matrices.  Returns what the
    CombinedGreensIterator measurements push: CDS, SDSx/y/z per direction, PS[dir12, dir1, dir2]."""
    N = L * L
    dirs, table = square_pair_directions(L)
    nd = len(dirs)
    trg = [[[t for t in range(N) if table[s, t] == k] for k in range(K)] for s in range(N)]
    out = {k: np.zeros(nd) for k in ("CDS", "SDSx", "SDSy", "SDSz")}
    out["PS"] = np.zeros((nd, K, K))
    G00 = full_greens(g00_blocks)
    for g0l, gl0, gll in steps:
        pg = (G00, full_greens(g0l), full_greens(gl0), full_greens(gll))
        for i in range(N):
            for j in range(N):
                d = table[i, j]
                if attractive:  # HubbardModelAttractive.jl:226-241
                    x = pg[1][j, i] * pg[2][i, j]
                    out["CDS"][d] += 4 * (1 - pg[3][i, i]) * (1 - pg[0][j, j]) - 2 * x
                    out["SDSx"][d] += -2 * x; out["SDSy"][d] += -2 * x; out["SDSz"][d] += -2 * x
                else:
                    out["CDS"][d] += cdc_kernel_packed(pg, N, i, j)
                    out["SDSx"][d] += sdc_x_kernel_packed(pg, N, i, j)
                    out["SDSy"][d] += sdc_y_kernel_packed(pg, N, i, j)
                    out["SDSz"][d] += sdc_z_kernel_packed(pg, N, i, j)
                for k1 in range(K):
                    for t1 in trg[i][k1]:
                        for k2 in range(K):
                            for t2 in trg[j][k2]:
                                if attractive:  # HubbardModelAttractive.jl:246-248
                                    out["PS"][d, k1, k2] += pg[2][i, j] * pg[2][t1, t2]
                                else:
                                    out["PS"][d, k1, k2] += pc_kernel_packed(pg, N, i, t1, j, t2)
    for k in out:
        out[k] = out[k] * delta_tau / N
    return out
