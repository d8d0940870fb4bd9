"""Hubbard models as the DQMC stack sees them (src/models/HubbardModel/*.jl).

The structs carry exactly the fields the hot path reads (U, t, mu, l, flv); the
performance-critical methods (interaction_matrix_exp!, propose_local,
accept_local!) live on the device behind dqmc_sweep_spatial / dqmc_propagate."""
import numpy as np

from .lattices import Chain, SquareLattice


def choose_lattice(dims, L):
    """HubbardModel.jl:23-31 (CubicLattice is outside the hot-path scope)."""
    if dims == 1:
        return Chain(L)
    if dims == 2:
        return SquareLattice(L)
    raise NotImplementedError("only Chain and SquareLattice are provided")


class HubbardModelAttractive:
    """HubbardModelAttractive.jl:26-39; U is stored positive."""
    flv = 1
    kind = 0

    def __init__(self, L=None, dims=None, l=None, U=1.0, t=1.0, mu=0.0):
        if U < 0:
            raise ValueError("U must be positive.")  # @assert U >= 0.
        self.l = l if l is not None else choose_lattice(dims, L)
        self.U, self.t, self.mu = float(U), float(t), float(mu)

    def hopping_matrix(self):
        """HubbardModelAttractive.jl:78-91"""
        N = len(self.l)
        T = np.diag(np.full(N, -self.mu))
        for src, trg in self.l.neighbors(True):
            T[trg - 1, src - 1] += -self.t
        return [T]


class HubbardModelRepulsive:
    """HubbardModelRepulsive.jl:24-41; two spin blocks (BlockDiagonal, :68-69)."""
    flv = 2
    kind = 1

    def __init__(self, L=None, dims=None, l=None, U=1.0, t=1.0):
        if U < 0:
            raise ValueError("U must be positive.")
        self.l = l if l is not None else choose_lattice(dims, L)
        self.U, self.t, self.mu = float(U), float(t), 0.0

    def hopping_matrix(self):
        """HubbardModelRepulsive.jl:87-100: BlockDiagonal(T, copy(T))"""
        N = len(self.l)
        T = np.zeros((N, N))
        for src, trg in self.l.neighbors(True):
            T[trg - 1, src - 1] += -self.t
        return [T, T.copy()]


def HubbardModel(*args, U, **kwargs):
    """HubbardModel.jl:14-20: U > 0 repulsive, else attractive with U -> -U"""
    if U > 0.0:
        return HubbardModelRepulsive(*args, U=U, **kwargs)
    return HubbardModelAttractive(*args, U=-U, **kwargs)


def rand_conf(rng, n_sites, slices):
    """rand(DQMC, m, nslices) (HubbardModel.jl:46-48): Int8 ±1, filled column-major."""
    flat = rng.integers(0, 2, size=n_sites * slices, dtype=np.int8) * 2 - 1
    return np.asfortranarray(flat.reshape((n_sites, slices), order="F").astype(np.int8))
