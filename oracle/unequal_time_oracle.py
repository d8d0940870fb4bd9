"""CPU restatement of MonteCarlo.jl's unequal-time Green's functions. TEST INFRASTRUCTURE ONLY.

Follows src/flavors/DQMC/unequal_time_stack.jl step by step (line numbers cited per function),
per block (the two blocks of the repulsive model are independent problems), on top of the C
oracle's primitives (udt_AVX_pivot!, rdivp!, slice matrices: oracle/dqmc_oracle.c).  Only tests/
may import this module.

Pinned by (a) the properties the reference's own test asserts (test/flavortests_DQMC.jl:75-162:
forward/backward stacks equal the DQMC stack, G(k,k) equals calculate_greens(k), G(t,0) + G(0,beta-t)
= 0, iterators equal greens(mc,k,l)) and (b) `brute_force_greens`, a dense evaluation of the
definition G(k,l) = B_k...B_{l+1} G(l,l) (k >= l), -(1 - G(k,k)) (B_l...B_{k+1})^-1 (k < l) that shares
no code with the stabilised path.
"""
import numpy as np

from . import oracle as O


def _I(n):
    return np.eye(n)


class UnequalTimeOracle:
    """UnequalTimeStack + calculate_greens(mc, k, l) + iterators for ONE block of an OracleDQMC"""

    def __init__(self, mc, block=0):
        self.mc, self.b = mc, block
        self.n = mc.N
        self.M = mc.slices
        self.s = mc.safe_mult
        self.nr = self.M // self.s  # length(mc.s.ranges)
        self.initialize_stack()

    # slice matrices of this block (slice_matrices.jl:10-39), 1-based slice
    def B(self, l):
        return self.mc.slice_matrix(l, 1.0)[self.b]

    def Binv(self, l):
        return self.mc.slice_matrix(l, -1.0)[self.b]

    def ranges(self, idx):  # mc.s.ranges[idx], idx 1-based (stack.jl:119)
        return range((idx - 1) * self.s + 1, idx * self.s + 1)

    @staticmethod
    def udt(X, apply_pivot=True):
        U, D, T, piv = O.udt_pivot(X, apply_pivot)
        return U, D, T, piv

    # unequal_time_stack.jl:54-100
    def initialize_stack(self):
        n, m = self.n, self.nr + 1
        self.fu = [_I(n) for _ in range(m)]; self.fd = [np.ones(n) for _ in range(m)]; self.ft = [_I(n) for _ in range(m)]
        self.bu = [_I(n) for _ in range(m)]; self.bd = [np.ones(n) for _ in range(m)]; self.bt = [_I(n) for _ in range(m)]
        self.iu = [None] * self.nr; self.id = [None] * self.nr; self.it = [None] * self.nr
        self.inv_done = [False] * self.nr
        self.forward_idx = 1
        self.backward_idx = self.nr + 1
        self.U, self.D, self.T = _I(n), np.ones(n), _I(n)

    def invalidate(self):
        """what `s.last_update != mc.last_sweep` triggers (unequal_time_stack.jl:164-169)"""
        self.inv_done = [False] * self.nr
        self.forward_idx = 1
        self.backward_idx = self.nr + 1

    # :162-190 (1-based idx as in the reference; lists are 0-based)
    def lazy_build_forward(self, upto):
        for idx in range(self.forward_idx, upto):
            cur = self.fu[idx - 1].copy()
            for sl in self.ranges(idx):
                cur = self.B(sl) @ cur
            tmp = cur * self.fd[idx - 1][None, :]
            self.fu[idx], self.fd[idx], t1, _ = self.udt(tmp)
            self.ft[idx] = t1 @ self.ft[idx - 1]
        self.forward_idx = max(upto, self.forward_idx)

    # :192-218
    def lazy_build_backward(self, downto):
        for idx in range(self.backward_idx - 1, downto - 1, -1):
            cur = self.bu[idx].copy()
            for sl in reversed(self.ranges(idx)):
                cur = self.B(sl).T @ cur
            tmp = cur * self.bd[idx][None, :]
            self.bu[idx - 1], self.bd[idx - 1], t1, _ = self.udt(tmp)
            self.bt[idx - 1] = t1 @ self.bt[idx]
        self.backward_idx = min(downto, self.backward_idx)

    # :220-246
    def lazy_build_inv(self, frm, to):
        for idx in range(frm, to + 1):
            if self.inv_done[idx - 1]:
                continue
            self.inv_done[idx - 1] = True
            t = _I(self.n)
            for sl in reversed(self.ranges(idx)):
                t = self.Binv(sl) @ t
            self.iu[idx - 1], self.id[idx - 1], self.it[idx - 1], _ = self.udt(t)

    # :106-160
    def build_stack(self):
        self.lazy_build_forward(self.nr + 1)
        self.lazy_build_backward(1)
        self.lazy_build_inv(1, self.nr)

    # :322-378  U D T = B_{low+1}^-1 ... B_high^-1
    def compute_inverse_udt_block(self, low, high):
        s, n = self.s, self.n
        lower = (low + 1 + s - 2) // s + 1
        upper = high // s
        self.lazy_build_inv(lower, upper)
        U, D, T = _I(n), np.ones(n), _I(n)
        for idx in range(lower, upper + 1):
            tmp1 = T @ self.iu[idx - 1]
            tmp2 = D[:, None] * tmp1
            tmp1 = tmp2 * self.id[idx - 1][None, :]
            tmp2, D, tmp1, _ = self.udt(tmp1)
            T = tmp1 @ self.it[idx - 1]
            U = U @ tmp2
        lower_slice = (lower - 1) * s + 1
        upper_slice = upper * s
        top = min(lower_slice - 1, high)
        for sl in range(top, low, -1):
            U = self.Binv(sl) @ U
        if top >= low + 1:
            tmp1 = U * D[None, :]
            U, D, tmp1, _ = self.udt(tmp1)
            T = tmp1 @ T
        for sl in range(max(upper_slice + 1, top + 1), high + 1):
            T = T @ self.Binv(sl)
        self.U, self.D, self.T = U, D, T

    # :392-410  Ul Dl Tl = B_slice ... B_1
    def compute_forward_udt_block(self, slice_):
        s = self.s
        idx = (slice_ - 1) // s if slice_ >= 1 else -1 // s  # Julia div truncates toward zero
        if slice_ < 1:
            idx = 0
        self.lazy_build_forward(idx + 1)
        T = self.fu[idx].copy()
        for l in range(s * idx + 1, slice_ + 1):
            T = self.B(l) @ T
        tmp = T * self.fd[idx][None, :]
        Ul, Dl, tmp, _ = self.udt(tmp)
        return Ul, Dl, tmp @ self.ft[idx]

    # :425-443  (Ur Dr Tr)^dagger = B_M ... B_{slice+1}
    def compute_backward_udt_block(self, slice_):
        s = self.s
        idx = (slice_ + s - 1) // s
        self.lazy_build_backward(idx + 1)
        U = self.bu[idx].copy()
        for l in range(s * idx, slice_, -1):
            U = self.B(l).T @ U
        tmp = U * self.bd[idx][None, :]
        Ur, Dr, tmp, _ = self.udt(tmp)
        return Ur, Dr, tmp @ self.bt[idx]

    # :447-530
    def calculate_greens_full1(self, slice1, slice2):
        self.compute_inverse_udt_block(slice2, slice1)
        Ul, Dl, Tl = self.compute_forward_udt_block(slice2)
        Ur, Dr, Tr = self.compute_backward_udt_block(slice1)
        U, D, T = self.U, self.D, self.T
        vmin = lambda w: np.minimum(1.0, w)
        vmaxinv = lambda w: 1.0 / np.maximum(1.0, w)
        g = Tl @ Tr.T                                   # B1
        g = Dl[:, None] * (g * Dr[None, :])
        Tr, Dr, g, piv = self.udt(g, False)             # udt
        Tl = Ul @ Tr                                    # B2
        Ur = O.rdivp(Ur, g, piv)
        Tr = U.T @ Tl                                   # B3
        Tr = (vmaxinv(D)[:, None] * Tr) * vmin(Dr)[None, :]
        Tl = T @ Ur                                     # B4
        Tl = (vmin(D)[:, None] * Tl) * vmaxinv(Dr)[None, :]
        Tl = Tl + Tr                                    # sum, UDT
        Tr, Dl, Tl, piv = self.udt(Tl, False)
        Dr = vmaxinv(Dr)                                # B5
        Ul = O.rdivp(np.diag(Dr), Tl, piv)
        Ul = (Ul * (1.0 / Dl)[None, :]) @ Tr.T
        g = Ul * vmaxinv(D)[None, :]
        g = Ur @ (g @ U.T)                              # B6
        return g

    # :534-605
    def calculate_greens_full2(self, slice1, slice2):
        self.compute_inverse_udt_block(slice1, slice2)
        Ul, Dl, Tl = self.compute_forward_udt_block(slice1)
        Ur, Dr, Tr = self.compute_backward_udt_block(slice2)
        U, D, T = self.U, self.D, self.T
        vmin = lambda w: np.minimum(1.0, w)
        vmaxinv = lambda w: 1.0 / np.maximum(1.0, w)
        g = Tl @ Tr.T                                   # B1
        g = (Dl[:, None] * g) * Dr[None, :]
        Tr, Dr, g, piv = self.udt(g, False)             # udt
        Tl = Ul @ Tr                                    # B2
        Ul = U.T @ Tl
        Ul = (vmaxinv(D)[:, None] * Ul) * vmin(Dr)[None, :]
        Us = T @ Ur                                     # B3
        Us = O.rdivp(Us, g, piv)
        Tr = (vmin(D)[:, None] * Us) * vmaxinv(Dr)[None, :]
        Tr = Tr + Ul                                    # sum, udt
        Ul, Dl, Tr, piv = self.udt(Tr, False)
        Us = O.rdivp(np.diag(vmin(Dr)), Tr, piv)        # B4
        Us = (Us * (1.0 / Dl)[None, :]) @ Ul.T
        Ur = Us * vmin(D)[None, :]
        g = -(Tl @ (Ur @ T))                            # B6
        return g

    # :288-303
    def calculate_greens(self, slice1, slice2):
        assert 0 <= slice1 <= self.M and 0 <= slice2 <= self.M
        if slice1 >= slice2:
            return self.calculate_greens_full1(slice1, slice2)
        return self.calculate_greens_full2(slice1, slice2)

    # _greens! (DQMC.jl:721-730)
    def to_true(self, g):
        return self.mc.eTinv @ (g @ self.mc.eT)

    def greens(self, slice1, slice2):
        return self.to_true(self.calculate_greens(slice1, slice2))

    # GreensIterator{:, l} (:644-715): yields greens(k, l) for k = l..M
    def greens_iterator(self, l=0, recalculate=None):
        recalculate = 4 * self.s if recalculate is None else recalculate
        g = self.calculate_greens_full1(l, l)
        U, D, T, _ = self.udt(g)
        yield self.to_true(g)
        for k in range(l + 1, self.M + 1):
            if k % recalculate == 0:
                g = self.calculate_greens_full1(k, l)
                out = self.to_true(g)
                U, D, T, _ = self.udt(g)
                yield out
            elif k % self.s == 0:
                U = self.B(k) @ U
                cur = U * D[None, :]
                tmp1 = cur @ T
                U, D, cur, _ = self.udt(cur)
                T = cur @ T
                yield self.to_true(tmp1)
            else:
                U = self.B(k) @ U
                yield self.to_true((U * D[None, :]) @ T)

    # CombinedGreensIterator (:746-883): yields (G0l, Gl0, Gll) for l = 1..M, starting from the
    # equal-time effective Green's function at current_slice == 1 (tau = 0)
    def combined_greens_iterator(self, greens_eff, recalculate=None):
        recalculate = 4 * self.s if recalculate is None else recalculate
        self.build_stack()
        Ul, Dl, Tl, _ = self.udt(greens_eff)
        uU, uD, uT = Ul.copy(), Dl.copy(), Tl.copy()
        Ur, Dr, Tr, _ = self.udt(greens_eff - _I(self.n))
        for l in range(1, self.M + 1):
            if l % recalculate == 0:
                gl0 = self.calculate_greens_full1(l, 0)
                g0l = self.calculate_greens_full2(0, l)
                gll = self.calculate_greens_full1(l, l)
                out = (self.to_true(g0l), self.to_true(gl0), self.to_true(gll))
                uU, uD, uT, _ = self.udt(gll)
                Ul, Dl, Tl, _ = self.udt(gl0)
                Ur, Dr, Tr, _ = self.udt(g0l)
                yield out
            elif l % self.s == 0:
                Ul = self.B(l) @ Ul
                Tr = Tr @ self.Binv(l)
                uU = self.B(l) @ uU
                uT = uT @ self.Binv(l)
                tmp1 = Ul * Dl[None, :]                 # Gl0
                tmp2 = tmp1 @ Tl
                Ul, Dl, tmp1, _ = self.udt(tmp1)
                Tl = tmp1 @ Tl
                gl0 = self.to_true(tmp2)
                cur = Dr[:, None] * Tr                  # G0l
                g0l_eff = Ur @ cur
                t2, Dr, Tr, _ = self.udt(cur)
                Ur = Ur @ t2
                g0l = self.to_true(g0l_eff)
                utmp = uU * uD[None, :]                 # Gll
                gll_eff = utmp @ uT
                cur, uD, utmp, _ = self.udt(utmp)
                uU2 = utmp @ uT
                uT2 = uD[:, None] * uU2
                utmp, uD, uT, _ = self.udt(uT2)
                uU = cur @ utmp
                yield (g0l, gl0, self.to_true(gll_eff))
            else:
                Ul = self.B(l) @ Ul
                Tr = Tr @ self.Binv(l)
                uU = self.B(l) @ uU
                uT = uT @ self.Binv(l)
                gl0 = self.to_true((Ul * Dl[None, :]) @ Tl)
                g0l = self.to_true((Ur * Dr[None, :]) @ Tr)
                gll = self.to_true((uU * uD[None, :]) @ uT)
                yield (g0l, gl0, gll)


def brute_force_greens(mc, block, k, l):
    """effective G(k, l) from its definition, dense linear algebra only:
    G(l,l) = [1 + B_l...B_1 B_M...B_{l+1}]^-1,  G(k,l) = B_k...B_{l+1} G(l,l) for k >= l,
    G(k,l) = -(1 - G(k,k)) (B_l...B_{k+1})^-1 for k < l.  Usable for small beta only."""
    n, M = mc.N, mc.slices
    B = lambda s: mc.slice_matrix(s, 1.0)[block]

    def prod(hi, lo):  # B_hi ... B_lo
        P = np.eye(n)
        for s in range(lo, hi + 1):
            P = B(s) @ P
        return P

    def equal(t):
        return np.linalg.inv(np.eye(n) + prod(t, 1) @ prod(M, t + 1))

    if k >= l:
        return prod(k, l + 1) @ equal(l)
    return -(np.eye(n) - equal(k)) @ np.linalg.inv(prod(l, k + 1))
