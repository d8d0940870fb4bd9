"""ctypes binding of the CPU oracle (oracle/dqmc_oracle.c). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (montecarlo.jl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdqmc_oracle.so")

ATTRACTIVE, REPULSIVE = 0, 1


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dqmc_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class MagStats(C.Structure):
    _fields_ = [("max", C.c_double), ("min", C.c_double), ("sum", C.c_double), ("count", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [
        ("prop_local", C.c_int64),
        ("acc_local", C.c_int64),
        ("imaginary_probability", MagStats),
        ("negative_probability", MagStats),
        ("propagation_error", MagStats),
    ]


class IsingResult(C.Structure):
    _fields_ = [("E", C.c_double), ("E2", C.c_double), ("M", C.c_double), ("M2", C.c_double),
                ("n_meas", C.c_int64), ("accepted", C.c_int64), ("proposed", C.c_int64)]


_lib = None


def lib(path=None):
    global _lib
    if _lib is None or path is not None:
        p = path or build()
        L = C.CDLL(p)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int64)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                 dp, dp, dp, dp, C.c_int, C.c_int]
        for name in ("orc_destroy", "orc_init_stack", "orc_build_stack", "orc_propagate",
                     "orc_sweep_spatial", "orc_update", "orc_prepare"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.orc_set_udt_presort.argtypes = [C.c_int]
        L.orc_set_udt_presort.restype = None
        L.orc_sweeps.argtypes = [C.c_void_p, C.c_int]
        L.orc_sweeps.restype = None
        L.orc_update_until_measure.argtypes = [C.c_void_p]
        L.orc_update_until_measure.restype = C.c_int
        L.orc_current_slice.argtypes = [C.c_void_p]
        L.orc_direction.argtypes = [C.c_void_p]
        L.orc_nblocks.argtypes = [C.c_void_p]
        L.orc_set_conf.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_conf.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_uniforms.argtypes = [C.c_void_p, dp, C.c_size_t]
        L.orc_uniforms_used.argtypes = [C.c_void_p]
        L.orc_uniforms_used.restype = C.c_size_t
        L.orc_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_get_greens_eff.argtypes = [C.c_void_p, dp]
        L.orc_set_greens_eff.argtypes = [C.c_void_p, dp]
        L.orc_get_greens.argtypes = [C.c_void_p, dp]
        L.orc_calculate_greens_at.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_wrap_greens.argtypes = [C.c_void_p, dp, C.c_int, C.c_int]
        L.orc_slice_matrix.argtypes = [C.c_void_p, C.c_int, C.c_double, dp]
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.orc_philox_uniform.restype = C.c_double
        L.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_square_neighs.argtypes = [C.c_int, ip]
        L.orc_square_bonds.argtypes = [C.c_int, ip]
        L.orc_build_checkerboard.argtypes = [C.c_int, C.c_int, ip, ip, ip, ip, C.c_int]
        L.orc_build_checkerboard.restype = C.c_int
        L.orc_hopping_square.argtypes = [C.c_int, C.c_double, C.c_double, dp]
        for name in ("orc_vmul_nn", "orc_vmul_nt", "orc_vmul_tn", "orc_vmul_tt", "orc_vmul_nd", "orc_vmul_dn"):
            getattr(L, name).argtypes = [C.c_int, dp, dp, dp]
            getattr(L, name).restype = None
        L.orc_set_dgemm_hook.argtypes = [C.c_void_p]
        L.orc_set_dgemm_hook.restype = None
        L.orc_rdivp.argtypes = [C.c_int, dp, dp, dp, ip]
        L.orc_udt_pivot.argtypes = [C.c_int, dp, dp, dp, ip, dp, C.c_int]
        L.orc_calculate_greens.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, dp]
        L.orc_ising_run.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_void_p,
                                    C.POINTER(IsingResult)]
        _lib = L
    return _lib


def use_openblas_dgemm(on=True):
    """Timing-only (bench.py "strong CPU" leg): route the oracle's four dense products to the dgemm of the BLAS that
    scipy links (OpenBLAS in this image), through scipy.linalg.cython_blas's C-API capsule.  Returns False if the
    capsule is not available.  Never used by the parity tests."""
    if not on:
        lib().orc_set_dgemm_hook(None)
        return True
    try:
        import scipy.linalg.cython_blas as cb
        cap = cb.__pyx_capi__["dgemm"]
        C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
        C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
        C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
        C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
        ptr = C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap))
    except Exception:
        return False
    lib().orc_set_dgemm_hook(ptr)
    return True


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def F(a):
    """column-major float64 copy"""
    return np.array(a, dtype=np.float64, order="F")


# ---------------------------------------------------------------- lattice
def square_neighs(L):
    out = np.zeros((4, L * L), dtype=np.int64, order="F")
    lib().orc_square_neighs(L, _ip(out))
    return out


def square_bonds(L):
    out = np.zeros((2 * L * L, 3), dtype=np.int64, order="F")
    lib().orc_square_bonds(L, _ip(out))
    return out


def build_checkerboard(n_sites, bonds):
    nb = bonds.shape[0]
    b = np.array(bonds, dtype=np.int64, order="F")
    cb = np.zeros((3, nb), dtype=np.int64, order="F")
    gs = np.zeros(64, dtype=np.int64)
    ge = np.zeros(64, dtype=np.int64)
    ng = lib().orc_build_checkerboard(n_sites, nb, _ip(b), _ip(cb), _ip(gs), _ip(ge), 64)
    return cb, [(int(gs[i]), int(ge[i])) for i in range(ng)], ng


def hopping_square(L, t=1.0, mu=0.0):
    T = np.zeros((L * L, L * L), order="F")
    lib().orc_hopping_square(L, t, mu, _dp(T))
    return T


def hopping_exponentials(T, dtau):
    """stack.jl:167-181.  Julia's exp(::Matrix) takes the Hermitian eigen path for
    a symmetric argument: V*Diagonal(exp.(w))*V'."""
    w, V = np.linalg.eigh(-0.5 * dtau * T)
    eT = (V * np.exp(w)) @ V.T
    w2, V2 = np.linalg.eigh(0.5 * dtau * T)
    eTinv = (V2 * np.exp(w2)) @ V2.T
    return F(eT), F(eTinv), F(eT @ eT), F(eTinv @ eTinv)


# ---------------------------------------------------------------- linalg
def vmul(kind, A, B):
    n = A.shape[0]
    A, B = F(A), F(B)
    Cm = np.zeros((n, n), order="F")
    getattr(lib(), "orc_vmul_" + kind)(n, _dp(Cm), _dp(A), _dp(B))
    return Cm


def udt_pivot(X, apply_pivot=True):
    n = X.shape[0]
    T = F(X)
    U = np.zeros((n, n), order="F")
    D = np.zeros(n)
    piv = np.zeros(n, dtype=np.int64)
    tmp = np.zeros(n)
    lib().orc_udt_pivot(n, _dp(U), _dp(D), _dp(T), _ip(piv), _dp(tmp), 1 if apply_pivot else 0)
    return U, D, T, piv


def rdivp(A, T, pivot):
    n = A.shape[0]
    A = F(A)
    T = F(T)
    O = np.zeros((n, n), order="F")
    piv = np.array(pivot, dtype=np.int64)
    lib().orc_rdivp(n, _dp(A), _dp(T), _dp(O), _ip(piv))
    return A


def calculate_greens(Ul, Dl, Tl, Ur, Dr, Tr):
    n = Ul.shape[0]
    args = [F(Ul), np.array(Dl, dtype=np.float64), F(Tl), F(Ur), np.array(Dr, dtype=np.float64), F(Tr)]
    G = np.zeros((n, n), order="F")
    piv = np.zeros(n, dtype=np.int64)
    tmp = np.zeros(n)
    lib().orc_calculate_greens(n, *[_dp(a) for a in args], _dp(G), _ip(piv), _dp(tmp))
    return G


def philox_uniform(seed, index):
    return lib().orc_philox_uniform(seed, index)


def philox_uniforms(seed, n, start=0):
    f = lib().orc_philox_uniform
    return np.array([f(seed, start + i) for i in range(n)])


def random_conf(seed, n_sites, slices):
    """Initial HS field: i.i.d. ±1 (HubbardModel.jl:46-48), drawn column-major from a
    numpy Philox stream keyed by `seed` (Julia's MersenneTwister is not reproducible here)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    return np.asfortranarray((2 * rng.integers(0, 2, size=(slices, n_sites)).T - 1).astype(np.int8))


# ---------------------------------------------------------------- DQMC object
class OracleDQMC:
    """Mirror of the reference DQMC object for one chain (DQMC.jl:133-189)."""

    def __init__(self, L, model="attractive", beta=1.0, delta_tau=0.1, safe_mult=10, U=1.0, t=1.0,
                 mu=0.0, check_propagation_error=True, check_sign_problem=True, hopping=None, exps=None):
        self.L = L
        self.N = L * L if hopping is None else hopping.shape[0]
        self.model = ATTRACTIVE if model == "attractive" else REPULSIVE
        self.nb = 1 if self.model == ATTRACTIVE else 2
        self.slices = int(round(beta / delta_tau))  # DQMC.jl:109
        self.delta_tau = delta_tau
        self.safe_mult = safe_mult
        self.U = U
        if hopping is None:
            hopping = hopping_square(L, t, mu if self.model == ATTRACTIVE else 0.0)
        self.T = hopping
        # exps: the four constant matrices handed over explicitly (e.g. the checkerboard group products)
        eT, eTinv, eT2, eTinv2 = [F(e) for e in exps] if exps is not None else hopping_exponentials(self.T, delta_tau)
        self.eT, self.eTinv, self.eT2, self.eTinv2 = eT, eTinv, eT2, eTinv2
        rep = lambda a: F(np.concatenate([a.reshape(-1, order="F")] * self.nb))
        self._c = [rep(eT), rep(eTinv), rep(eT2), rep(eTinv2)]
        self.h = lib().orc_create(self.N, self.model, self.slices, safe_mult, delta_tau, U,
                                  *[_dp(a) for a in self._c], int(check_propagation_error),
                                  int(check_sign_problem))
        if not self.h:
            raise ValueError("slices must be divisible by safe_mult (stack.jl:115)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    # state
    def set_conf(self, conf):
        c = np.array(conf, dtype=np.int8, order="F")
        assert c.shape == (self.N, self.slices)
        lib().orc_set_conf(self.h, c.ctypes.data)

    def conf(self):
        c = np.zeros((self.N, self.slices), dtype=np.int8, order="F")
        lib().orc_get_conf(self.h, c.ctypes.data)
        return c

    def set_uniforms(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        lib().orc_set_uniforms(self.h, _dp(u), u.size)

    def uniforms_used(self):
        return lib().orc_uniforms_used(self.h)

    def seed(self, s):
        lib().orc_seed(self.h, s)

    # driver
    def init_stack(self): lib().orc_init_stack(self.h)
    def build_stack(self): lib().orc_build_stack(self.h)
    def propagate(self): lib().orc_propagate(self.h)
    def sweep_spatial(self): lib().orc_sweep_spatial(self.h)
    def update(self): lib().orc_update(self.h)
    def prepare(self): lib().orc_prepare(self.h)
    def sweeps(self, n): lib().orc_sweeps(self.h, n)
    def update_until_measure(self): return lib().orc_update_until_measure(self.h)

    @property
    def current_slice(self): return lib().orc_current_slice(self.h)
    @property
    def direction(self): return lib().orc_direction(self.h)

    def _blocks(self, flat):
        n = self.N
        return [flat[b * n * n:(b + 1) * n * n].reshape((n, n), order="F") for b in range(self.nb)]

    def greens_eff(self):
        out = np.zeros(self.nb * self.N * self.N)
        lib().orc_get_greens_eff(self.h, _dp(out))
        return self._blocks(out)

    def set_greens_eff(self, blocks):
        flat = np.concatenate([F(b).reshape(-1, order="F") for b in blocks])
        lib().orc_set_greens_eff(self.h, _dp(flat))

    def greens(self):
        out = np.zeros(self.nb * self.N * self.N)
        lib().orc_get_greens(self.h, _dp(out))
        return self._blocks(out)

    def calculate_greens_at(self, slice_):
        out = np.zeros(self.nb * self.N * self.N)
        lib().orc_calculate_greens_at(self.h, slice_, _dp(out))
        return self._blocks(out)

    def wrap_greens(self, blocks, slice_, direction):
        flat = np.concatenate([F(b).reshape(-1, order="F") for b in blocks])
        lib().orc_wrap_greens(self.h, _dp(flat), slice_, direction)
        return self._blocks(flat)

    def slice_matrix(self, slice_, power=1.0):
        out = np.zeros(self.nb * self.N * self.N)
        lib().orc_slice_matrix(self.h, slice_, power, _dp(out))
        return self._blocks(out)

    def stats(self):
        st = Stats()
        lib().orc_get_stats(self.h, C.byref(st))
        return st


def ising_run(L, beta, thermalization, sweeps, seed):
    res = IsingResult()
    lib().orc_ising_run(L, beta, thermalization, sweeps, seed, None, C.byref(res))
    return res
