"""Host-side lattices (src/lattices/square.jl:25-60, chain.jl:19-41, abstract.jl:99-115).

Integer tables are 1-based like the reference's so that they compare verbatim with
its fixtures (test/flavortests_DQMC.jl:22-24)."""
import numpy as np


class AbstractLattice:
    def __len__(self):
        return self.sites

    def neighbors(self, directed=False):
        """neighbors(l, Val(directed)), src/lattices/abstract.jl:99-108"""
        if directed:  # src-major, then the rows of neighs
            return [(src + 1, int(trg)) for src in range(self.sites) for trg in self.neighs[:, src]]
        return [(int(b[0]), int(b[1])) for b in self.bonds]


class SquareLattice(AbstractLattice):
    """SquareLattice(L): neighs rows = up, right, down, left (square.jl:47-60)."""

    def __init__(self, L):
        self.L = L
        self.sites = L * L
        lat = np.arange(1, L * L + 1).reshape((L, L), order="F")
        self.lattice = lat
        up = np.roll(lat, -1, axis=0)       # circshift(lattice, (-1, 0))
        right = np.roll(lat, -1, axis=1)    # circshift(lattice, (0, -1))
        down = np.roll(lat, 1, axis=0)
        left = np.roll(lat, 1, axis=1)
        self.neighs = np.vstack([x.reshape(-1, order="F") for x in (up, right, down, left)]).astype(np.int64)
        self.n_bonds = 2 * self.sites
        bonds = np.zeros((self.n_bonds, 3), dtype=np.int64)
        b = 0
        for src in lat.reshape(-1, order="F"):
            bonds[b] = (src, self.neighs[0, src - 1], 0); b += 1
            bonds[b] = (src, self.neighs[1, src - 1], 0); b += 1
        self.bonds = bonds


class Chain(AbstractLattice):
    """Chain(nsites): neighs rows = right, left (chain.jl:36-41)."""

    def __init__(self, nsites):
        self.sites = nsites
        c = np.arange(1, nsites + 1)
        self.neighs = np.vstack([np.roll(c, -1), np.roll(c, 1)]).astype(np.int64)
        self.n_bonds = nsites
        self.bonds = np.array([(s, self.neighs[0, s - 1], 0) for s in c], dtype=np.int64)


def build_checkerboard(l):
    """src/flavors/DQMC/abstract.jl:23-54 (used here only to pin the bond tables)."""
    bonds = l.neighbors(False)
    n_bonds = len(bonds)
    edges_used = np.zeros(n_bonds, dtype=np.int64)
    cb = np.zeros((3, n_bonds), dtype=np.int64)
    groups = []
    gs = ge = 1
    while edges_used.min() == 0:
        sites_used = np.zeros(len(l), dtype=np.int64)
        for idx, (src, trg) in enumerate(bonds):
            if edges_used[idx] or sites_used[src - 1] or sites_used[trg - 1]:
                continue
            edges_used[idx] = sites_used[src - 1] = sites_used[trg - 1] = 1
            cb[:, ge - 1] = (src, trg, idx + 1)
            ge += 1
        groups.append((gs, ge - 1))
        gs = ge
    return cb, groups, len(groups)
