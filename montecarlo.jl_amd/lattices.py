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


# ---------------------------------------------------------------------------
# EachSitePairByDistance (src/lattices/lattice_iterators.jl:131-190)
def _positions(l):
    """positions(l) (square.jl:72, chain.jl:52): 1-based cartesian coordinates"""
    if isinstance(l, SquareLattice):
        return [np.array([i + 1.0, j + 1.0]) for j in range(l.L) for i in range(l.L)]
    return [np.array([i + 1.0]) for i in range(l.sites)]


def _lattice_vectors(l):
    if isinstance(l, SquareLattice):
        return [np.array([float(l.L), 0.0]), np.array([0.0, float(l.L)])]
    return [np.array([float(l.sites)])]


def generate_combinations(vs):
    """lattice_iterators.jl:137-143: all periodic images, in the reference's order"""
    out = [np.zeros(len(vs[0]))]
    for v in vs:
        out = [e - v for e in out] + out + [e + v for e in out]
    return out


def directed_norm(v, eps):
    """norm + eps * angle(v, e_x) (lattice_iterators.jl:146-155)"""
    ln = float(np.linalg.norm(v))
    if len(v) == 2 and ln > eps:
        angle = float(np.arccos(v[0] / ln))
        if v[1] < 0:
            angle = 2 * np.pi - angle
        return ln + eps * angle
    return ln


class EachSitePairByDistance:
    """Triplets (direction index, source, target) sorted by distance.  `pairs[d]` lists the 1-based
    (src, trg) pairs of direction d in the reference's order; `dir_of[src-1, trg-1]` is the 0-based
    direction index of a pair; `directions[d]` the displacement vector."""

    def __init__(self, lattice, eps=1e-6):
        pos = _positions(lattice)
        wrap = generate_combinations(_lattice_vectors(lattice))
        directions, bonds = [], []
        for origin in range(len(lattice)):
            for trg, p in enumerate(pos):
                d = pos[origin] - p + wrap[0]
                for v in wrap[1:]:
                    new_d = pos[origin] - p + v
                    if directed_norm(new_d, eps) + eps < directed_norm(d, eps):
                        d = new_d
                idx = next((k for k, dd in enumerate(directions) if np.linalg.norm(dd - d) <= eps), None)
                if idx is None:
                    directions.append(d.copy())
                    bonds.append([])
                    idx = len(directions) - 1
                bonds[idx].append((origin + 1, trg + 1))
        order = sorted(range(len(directions)), key=lambda k: directed_norm(directions[k], eps))  # stable
        self.directions = [directions[k] for k in order]
        self.pairs = [bonds[k] for k in order]
        self.N = len(lattice) ** 2
        n = len(lattice)
        self.dir_of = np.zeros((n, n), dtype=np.int32)
        for d, prs in enumerate(self.pairs):
            for s, t in prs:
                self.dir_of[s - 1, t - 1] = d

    def __len__(self):
        return self.N

    def ndirections(self):
        return len(self.pairs)

    def __iter__(self):
        for d, prs in enumerate(self.pairs):
            for s, t in prs:
                yield d + 1, s, t


class EachLocalQuadByDistance:
    """EachLocalQuadByDistance{K}(lattice) (src/lattices/lattice_iterators.jl:258-353): quadruples
    (src1, trg1, src2, trg2) where trg_k is reached from src_k in one of the K shortest directions,
    grouped by (dir12, dir1, dir2) with dir12 the direction index of the pair (src1, src2).
    `trg_from_src[src-1]` lists the 1-based (dir, trg) of a source in the reference's order;
    `trg_of[src-1, k]` is the 0-based target in direction k (-1 if the site has none; sites of a
    Bravais lattice have exactly one per direction)."""

    def __init__(self, lattice, K=None, pairs=None):
        self.pairs_by_dir = pairs if pairs is not None else EachSitePairByDistance(lattice)
        n = len(lattice)
        if K is None:  # pairing(): K = 1 + length(neighbors(lattice, 1)) (measurements.jl:199-204)
            K = 1 + lattice.neighs.shape[0]
        if K > self.pairs_by_dir.ndirections():
            raise ValueError("K exceeds the number of directions of the lattice")
        self.K = K
        self.trg_from_src = [[] for _ in range(n)]
        for d in range(K):
            for src, trg in self.pairs_by_dir.pairs[d]:
                self.trg_from_src[src - 1].append((d + 1, trg))
        self.N = sum(len(x) for x in self.trg_from_src) ** 2
        self.trg_of = -np.ones((n, K), dtype=np.int32)
        for src, lst in enumerate(self.trg_from_src):
            for d, trg in lst:
                if self.trg_of[src, d - 1] >= 0:
                    raise ValueError("more than one target per direction: lattice with a basis is not supported")
                self.trg_of[src, d - 1] = trg - 1

    def __len__(self):
        return self.N

    def ndirections(self):
        return (self.pairs_by_dir.ndirections(), self.K, self.K)

    def __iter__(self):
        """(lin, src1, trg1, src2, trg2), lin the 1-based linear index of (dir12, dir1, dir2)"""
        nd, K = self.pairs_by_dir.ndirections(), self.K
        for d2 in range(K):
            for d1 in range(K):
                for d12 in range(nd):
                    lin = 1 + d12 + nd * (d1 + K * d2)
                    for src1, src2 in self.pairs_by_dir.pairs[d12]:
                        for dir1, trg1 in self.trg_from_src[src1 - 1]:
                            if dir1 != d1 + 1:
                                continue
                            for dir2, trg2 in self.trg_from_src[src2 - 1]:
                                if dir2 == d2 + 1:
                                    yield lin, src1, trg1, src2, trg2
