"""DQMC flavor for a batch of walkers on one MI355X (mirror of src/flavors/DQMC/DQMC.jl).

The object keeps the reference's names (`p`, `a`, `conf`, `propagate`, `sweep_spatial`,
`update`, `run`, `greens`, `current_slice`, ...) so that parity tests read like the
reference's tests; every numerical method is a call into libdqmc_hip.so.  One DQMC
object = `n_walkers` independent Markov chains advancing in lockstep."""
import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, dptr, i64ptr, lib
from .configurations import CompressedConf, ConfigRecorder
from .models import rand_conf


@dataclass
class DQMCParameters:
    """DQMC.jl:52-125"""
    thermalization: int = 100
    sweeps: int = 100
    check_sign_problem: bool = True
    check_propagation_error: bool = True
    safe_mult: int = 10
    delta_tau: float = 0.1
    beta: float = 1.0
    slices: int = 10
    measure_rate: int = 10

    @staticmethod
    def resolve(**kw):
        """the (beta, delta_tau, slices) resolution rules of DQMC.jl:84-110"""
        keys = set(k for k in ("beta", "delta_tau", "slices") if k in kw)
        if keys == {"beta"}:
            kw["delta_tau"] = 0.1
            keys.add("delta_tau")
        if keys == {"delta_tau", "beta", "slices"}:
            slices = int(round(kw["beta"] / kw["delta_tau"]))
            if slices != kw["slices"]:
                raise ValueError("Given slices (%d) does not match calculated slices beta/delta_tau ≈ %d"
                                 % (kw["slices"], slices))
        elif keys == {"beta", "slices"}:
            kw["delta_tau"] = kw["beta"] / kw["slices"]
        elif keys == {"delta_tau", "slices"}:
            kw["beta"] = kw["delta_tau"] * kw["slices"]
        elif keys == {"delta_tau", "beta"}:
            kw["slices"] = int(round(kw["beta"] / kw["delta_tau"]))
        else:
            raise ValueError("Invalid keyword arguments to DQMCParameters: %s" % sorted(keys))
        return DQMCParameters(**kw)


def hopping_exponentials(T, delta_tau):
    """init_hopping_matrix_exp (stack.jl:167-181).  Julia's exp(::Matrix) uses the
    Hermitian eigen-decomposition for a symmetric argument."""
    w, V = np.linalg.eigh(-0.5 * delta_tau * T)
    eT = (V * np.exp(w)) @ V.T
    w, V = np.linalg.eigh(0.5 * delta_tau * T)
    eTinv = (V * np.exp(w)) @ V.T
    return eT, eTinv, eT @ eT, eTinv @ eTinv


def checkerboard_exponentials(T, lattice, delta_tau, return_factors=False):
    """CheckerboardTrue (init_checkerboard_matrices, stack.jl:185-235; slice_matrices.jl:79-222;
    _greens! DQMC.jl:731-750) as the four constant matrices of the dense code path.  The reference
    keeps one sparse matrix per bond group, chkr_hop_half[g] = exp(-dtau/2 Tg) with
    Tg[trg, src] = T[trg, src] over the group's (disjoint) bonds, and applies
        B_l      = H_n ... H_2 C_1 H_2 ... H_n Mu eV_l,       C_1 = chkr_hop[1],  Mu = exp(-dtau diag(T))
        B_l^-1   = eV_l^-1 Mu^-1 (H_n ... H_2 C_1 H_2 ... H_n)^-1
        greens() = H_1^-1 ... H_n^-1  G  H_n ... H_1
    group by group.  On MI355X a pass of O(n^2 groups) sparse row updates is HBM-bound and slower than one
    MFMA GEMM at these sizes, so the group products are multiplied out once here (same order of factors)
    and handed to the same kernels: eT2 -> P Mu, eTinv2 -> Mu^-1 P^-1, eT -> H_n...H_1, eTinv -> its inverse."""
    from .lattices import build_checkerboard
    from scipy.linalg import expm
    N = T.shape[0]
    cb, groups, n_groups = build_checkerboard(lattice)

    def rem_eff_zeros(X):  # stack.jl:184
        X = X.copy()
        X[np.abs(X) < 1e-15] = 0.0
        return X

    H, Hinv, Cm, Cinv = [], [], [], []
    for (gs, ge) in groups:
        Tg = np.zeros((N, N))
        for i in range(gs, ge + 1):
            src, trg = cb[0, i - 1], cb[1, i - 1]
            Tg[trg - 1, src - 1] = T[trg - 1, src - 1]
        H.append(rem_eff_zeros(expm(-0.5 * delta_tau * Tg)))
        Hinv.append(rem_eff_zeros(expm(0.5 * delta_tau * Tg)))
        Cm.append(rem_eff_zeros(expm(-delta_tau * Tg)))
        Cinv.append(rem_eff_zeros(expm(delta_tau * Tg)))
    mus = np.diag(T)
    Mu, Muinv = np.diag(np.exp(-delta_tau * mus)), np.diag(np.exp(delta_tau * mus))

    if return_factors:
        return H, Hinv, Cm, Cinv, np.exp(-delta_tau * mus), np.exp(delta_tau * mus), n_groups

    def sandwich(half, full):  # the factor applied by multiply_slice_matrix_left! (slice_matrices.jl:109-121)
        M = np.eye(N)
        for i in reversed(range(1, n_groups)):
            M = half[i] @ M
        M = full[0] @ M
        for i in range(1, n_groups):
            M = half[i] @ M
        return M

    P, Pinv = sandwich(H, Cm), sandwich(Hinv, Cinv)
    eT = np.eye(N)
    for i in reversed(range(n_groups)):     # target * chkr_hop_half[i], i = n..1
        eT = eT @ H[i]
    eTinv = np.eye(N)
    for i in reversed(range(n_groups)):     # chkr_hop_half_inv[i] * target, i = n..1
        eTinv = Hinv[i] @ eTinv
    return eT, eTinv, P @ Mu, Muinv @ Pinv


def checkerboard_tables(T, lattice, delta_tau):
    """The sparse factors of init_checkerboard_matrices (stack.jl:185-235) in the ELL form dqmc_set_checkerboard takes:
    factor list = [H_1..H_n, C_1, Hinv_1..Hinv_n, Cinv_1] followed by their transposes (offset n_f), each as
    (vals, cols)[n_sites][kmax]; plus the seven sequences of include/dqmc_hip.h (0-based factor indices)."""
    H, Hinv, Cm, Cinv, mu, mu_inv, ng = checkerboard_exponentials(T, lattice, delta_tau, return_factors=True)
    mats = list(H) + [Cm[0]] + list(Hinv) + [Cinv[0]]
    nf = len(mats)
    mats = mats + [m.T for m in mats]
    N = T.shape[0]
    kmax = max(int((np.abs(m) > 0).sum(axis=1).max()) for m in mats)
    vals = np.zeros((len(mats), N, kmax))
    cols = np.zeros((len(mats), N, kmax), dtype=np.int32)
    for i, m in enumerate(mats):
        for r in range(N):
            nz = np.nonzero(m[r])[0]
            vals[i, r, :len(nz)] = m[r, nz]
            cols[i, r, :len(nz)] = nz
            cols[i, r, len(nz):] = r
    iH = lambda g: g              # H_g (g = 0..ng-1)
    iC = ng                       # C_1
    iHi = lambda g: ng + 1 + g    # Hinv_g
    iCi = 2 * ng + 1              # Cinv_1
    sand = lambda h, c: [h(g) for g in range(ng - 1, 0, -1)] + [c] + [h(g) for g in range(1, ng)]
    tr = lambda seq: [nf + i for i in seq]
    seqs = [sand(iH, iC),                                 # B X
            sand(iHi, iCi),                               # B^-1 X
            tr(sand(iH, iC)),                             # B' X
            tr(sand(iH, iC)),                             # X B  (columns mixed by rows of the transposes)
            tr(sand(iHi, iCi)),                           # X B^-1
            tr([iH(g) for g in range(ng - 1, -1, -1)]),   # X eT = X H_n ... H_1
            [iHi(g) for g in range(ng - 1, -1, -1)]]      # eTinv X = Hinv_1 ... Hinv_n X (Hinv_n applied first)
    return dict(kmax=kmax, vals=vals, cols=cols, mu=mu, mu_inv=mu_inv, seqs=seqs)


class DQMCAnalysis:
    """DQMC.jl:36-47 for one walker"""

    def __init__(self, st):
        self.prop_local = st.prop_local
        self.acc_local = st.acc_local
        self.acc_rate = st.acc_local / st.prop_local if st.prop_local else 0.0
        self.imaginary_probability = st.imaginary_probability
        self.negative_probability = st.negative_probability
        self.propagation_error = st.propagation_error


class DQMC:
    """DQMC(model; beta, delta_tau=0.1, safe_mult=10, ...) (DQMC.jl:250-289) for
    `n_walkers` chains on device `device_id`.  `seed` keys the initial HS fields and the
    Metropolis streams of walker w as `seed + first_walker + w`, so results do not depend
    on how walkers are distributed over devices."""

    def __init__(self, model, n_walkers=1, device_id=0, seed=123, first_walker=0, thermalization=100, sweeps=100,
                 safe_mult=10, measure_rate=10, check_sign_problem=True, check_propagation_error=True,
                 checkerboard=False, **kw):
        self.model = model
        self.checkerboard = bool(checkerboard)
        self.p = DQMCParameters.resolve(thermalization=thermalization, sweeps=sweeps, safe_mult=safe_mult,
                                        measure_rate=measure_rate, check_sign_problem=check_sign_problem,
                                        check_propagation_error=check_propagation_error, **kw)
        self.n_walkers = n_walkers
        self.N = len(model.l)
        self.nb = model.flv
        self.last_sweep = 0
        Ts = model.hopping_matrix()
        if self.checkerboard:  # DQMC(m; checkerboard=true) (DQMC.jl:250-263)
            exps = [checkerboard_exponentials(T, model.l, self.p.delta_tau) for T in Ts]
        else:
            exps = [hopping_exponentials(T, self.p.delta_tau) for T in Ts]
        cat = lambda k: np.ascontiguousarray(np.concatenate([e[k].reshape(-1, order="F") for e in exps]))
        self._eT, self._eTinv, self._eT2, self._eTinv2 = cat(0), cat(1), cat(2), cat(3)
        self.hopping_matrix_exp = [e[0] for e in exps]
        self.hopping_matrix_exp_inv = [e[1] for e in exps]
        self.hopping_matrix_exp_squared = [e[2] for e in exps]
        self.hopping_matrix_exp_inv_squared = [e[3] for e in exps]
        prm = _lib.Params(self.N, model.kind, self.p.slices, self.p.safe_mult, n_walkers, device_id,
                          int(self.p.check_propagation_error), int(self.p.check_sign_problem), self.p.delta_tau,
                          model.U, dptr(self._eT), dptr(self._eTinv), dptr(self._eT2), dptr(self._eTinv2))
        self._h = C.c_void_p()
        check(lib().dqmc_create(C.byref(prm), C.byref(self._h)))
        # checkerboard=True picks the faster execution of the same decomposition: measured on MI355X (config 3 shape,
        # tools/time_parts.py checkerboard) the sparse-factor kernel takes 32.5 us per product against 31.2 us for the dense
        # MFMA GEMM with the multiplied-out constants at n = 256, so dense up to n = 256 and sparse (O(n^2) work per
        # product) above; checkerboard="sparse" / "dense" force one
        sparse = checkerboard == "sparse" or (checkerboard is True and self.N > 256)
        if self.checkerboard and sparse:
            tb = [checkerboard_tables(T, model.l, self.p.delta_tau) for T in Ts]
            t0 = tb[0]  # both spin blocks of the repulsive model share T (HubbardModelRepulsive.jl:87-100)
            seqs = np.zeros((7, 32), dtype=np.int32)
            lens = np.zeros(7, dtype=np.int32)
            for q, sq in enumerate(t0["seqs"]):
                lens[q] = len(sq)
                seqs[q, :len(sq)] = sq
            mu = np.ascontiguousarray(np.concatenate([t["mu"] for t in tb]))
            mui = np.ascontiguousarray(np.concatenate([t["mu_inv"] for t in tb]))
            vals, cols = np.ascontiguousarray(t0["vals"]), np.ascontiguousarray(t0["cols"])
            check(lib().dqmc_set_checkerboard(self._h, t0["kmax"], vals.shape[0], dptr(vals),
                                              cols.ctypes.data_as(C.POINTER(C.c_int32)), dptr(mu), dptr(mui),
                                              seqs.ctypes.data_as(C.POINTER(C.c_int32)),
                                              lens.ctypes.data_as(C.POINTER(C.c_int32))), self._h)
        # rand(DQMC, m, slices) per walker (DQMC.jl:273), then the Metropolis stream
        self.seeds = [seed + first_walker + w for w in range(n_walkers)]
        for w, s in enumerate(self.seeds):
            rng = np.random.Generator(np.random.Philox(key=s))
            self.set_conf(w, rand_conf(rng, self.N, self.p.slices))
            self.seed(w, s)

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().dqmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self, rc):
        check(rc, self._h)

    # ---- state
    def set_conf(self, walker, conf):
        c = np.asfortranarray(np.asarray(conf, dtype=np.int8))
        if c.shape != (self.N, self.p.slices):
            raise ValueError("conf must have shape (n_sites, slices)")
        self._c(lib().dqmc_set_conf(self._h, walker, c.ctypes.data))

    def conf(self, walker=0):
        c = np.zeros((self.N, self.p.slices), dtype=np.int8, order="F")
        self._c(lib().dqmc_get_conf(self._h, walker, c.ctypes.data))
        return c

    def conf_bits(self, walker=0):
        """compress(mc, model, conf(mc)) packed on the device (HubbardModel.jl:56-59)"""
        n = (self.N * self.p.slices + 63) // 64
        ch = np.zeros(n, dtype=np.uint64)
        self._c(lib().dqmc_get_conf_bits(self._h, walker, ch.ctypes.data_as(C.POINTER(C.c_uint64))))
        return CompressedConf(ch, (self.N, self.p.slices))

    def set_conf_bits(self, walker, cc):
        """mc.conf = decompress(mc, model, c), unpacked on the device"""
        if cc.shape != (self.N, self.p.slices):
            raise ValueError("compressed configuration has the wrong shape")
        ch = np.ascontiguousarray(cc.chunks, dtype=np.uint64)
        self._c(lib().dqmc_set_conf_bits(self._h, walker, ch.ctypes.data_as(C.POINTER(C.c_uint64))))

    def seed(self, walker, seed):
        self._c(lib().dqmc_seed(self._h, walker, seed))

    def set_uniforms(self, walker, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        self._c(lib().dqmc_set_uniforms(self._h, walker, dptr(u), u.size))

    def uniforms_used(self, walker=0):
        n = C.c_uint64()
        self._c(lib().dqmc_uniforms_used(self._h, walker, C.byref(n)))
        return n.value

    def _state(self):
        cs, d = C.c_int32(), C.c_int32()
        self._c(lib().dqmc_get_state(self._h, C.byref(cs), C.byref(d)))
        return cs.value, d.value

    @property
    def current_slice(self):
        return self._state()[0]

    @property
    def direction(self):
        return self._state()[1]

    # ---- the sweep loop
    def prepare(self):
        """init!, build_stack, propagate (DQMC.jl:412-414)"""
        self._c(lib().dqmc_prepare(self._h))

    def build_stack(self):
        self._c(lib().dqmc_build_stack(self._h))

    def propagate(self):
        self._c(lib().dqmc_propagate(self._h))

    def sweep_spatial(self):
        self._c(lib().dqmc_sweep_spatial(self._h))

    def update(self):
        self._c(lib().dqmc_update(self._h))

    def sweep(self, n=1):
        self._c(lib().dqmc_sweep(self._h, n))

    def update_until_measure(self):
        n = C.c_int32()
        self._c(lib().dqmc_update_until_measure(self._h, C.byref(n)))
        return n.value

    def synchronize(self):
        self._c(lib().dqmc_synchronize(self._h))

    def replay_greens(self, slice_=0):
        """calculate_greens(mc, slice) into mc.s.greens for every walker (DQMC.jl:651-652)"""
        self._c(lib().dqmc_replay_greens(self._h, slice_))

    def replay(self, configurations, measure_rate=1):
        """replay!(mc, configurations) (DQMC.jl:605-697): the recorded configurations are decompressed
        on the device, `n_walkers` at a time, the Green's function is rebuilt from scratch at slice 0
        (calculate_greens(mc, 0), DQMC.jl:652) and the measurement sums are accumulated.
        Returns the accumulator vector (layout of include/dqmc_hip.h)."""
        cfgs = list(configurations)[::measure_rate]
        self.reset_accumulators()
        W, n, B = self.n_walkers, self.N, self.nb
        tail = np.zeros(self.accumulator_size())
        for i0 in range(0, len(cfgs), W):
            batch = cfgs[i0:i0 + W]
            for w in range(W):  # a short last batch repeats its last configuration in the idle walkers
                self.set_conf_bits(w, batch[min(w, len(batch) - 1)])
            self.replay_greens(0)
            if len(batch) == W:
                self.accumulate_greens()
            else:  # partial batch: only the filled walkers count
                for w in range(len(batch)):
                    G = self.greens(w)
                    g = np.concatenate([x.reshape(-1, order="F") for x in G])
                    tail[:B * n * n] += g
                    tail[B * n * n:2 * B * n * n] += g * g
                    for b in range(B):
                        tail[2 * B * n * n + b * n:2 * B * n * n + (b + 1) * n] += 1.0 - np.diag(G[b])
                    tail[-1] += 1
        return self.accumulators() + tail

    def run(self, verbose=False, on_measure=None, recorder=None, measurements=("greens",)):
        """run!(mc) (DQMC.jl:369-515) without the host-side measurement framework: the selected
        measurements are accumulated on the device every `measure_rate`-th sweep after thermalization,
        at current_slice == 1 && direction == +1 (DQMC.jl:425-436).  `measurements` may contain
        "greens" (greens_measurement, occupation), "correlations" (charge/spin density correlations,
        magnetization; needs set_pair_directions), "pairing" (needs set_local_targets) and
        "susceptibilities" (the CombinedGreensIterator measurements)."""
        known = {"greens": lib().dqmc_accumulate_greens, "correlations": lib().dqmc_accumulate_correlations,
                 "pairing": lib().dqmc_accumulate_pairing}
        for m in measurements:
            if m not in known and m != "susceptibilities":
                raise ValueError("unknown measurement %r" % (m,))
        self.prepare()
        if self.last_sweep == 0:  # a resumed run! keeps the measurement state (DQMC.jl:395-411)
            self.reset_accumulators()
        total = self.p.thermalization + self.p.sweeps
        t0 = time.time()
        for i in range(self.last_sweep + 1, total + 1):
            for _ in range(2 * self.p.slices):
                self._c(lib().dqmc_update(self._h))
                cs, d = self._state()
                if cs == 1 and d == 1 and i > self.p.thermalization and recorder is not None:
                    recorder.push(self, i)  # push!(mc.configs, mc, mc.model, i) (DQMC.jl:430)
                if cs == 1 and d == 1 and i > self.p.thermalization and i % self.p.measure_rate == 0:
                    for m in measurements:
                        if m == "susceptibilities":
                            self.accumulate_susceptibilities()
                        else:
                            self._c(known[m](self._h))
                    if on_measure is not None:
                        on_measure(self, i)
            self.last_sweep = i
            if verbose and i % 10 == 0:
                print("\t%d\n\t\tsweep dur: %.3fs" % (i, (time.time() - t0) / 10))
                t0 = time.time()
        self.synchronize()
        return True

    # ---- Green's functions
    def _blocks(self, flat):
        n = self.N
        return [flat[b * n * n:(b + 1) * n * n].reshape((n, n), order="F") for b in range(self.nb)]

    def greens_eff(self, walker=0):
        """mc.s.greens"""
        out = np.zeros(self.nb * self.N * self.N)
        self._c(lib().dqmc_get_greens_eff(self._h, walker, dptr(out)))
        return self._blocks(out)

    def set_greens_eff(self, walker, blocks):
        flat = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, order="F") for b in blocks]))
        self._c(lib().dqmc_set_greens_eff(self._h, walker, dptr(flat)))

    def greens(self, walker=0):
        """greens(mc) (DQMC.jl:711-730)"""
        out = np.zeros(self.nb * self.N * self.N)
        self._c(lib().dqmc_get_greens(self._h, walker, dptr(out)))
        return self._blocks(out)

    def calculate_greens(self, slice_, walker=0):
        """calculate_greens(mc, slice) (stack.jl:422-480)"""
        out = np.zeros(self.nb * self.N * self.N)
        self._c(lib().dqmc_calculate_greens_at(self._h, walker, slice_, dptr(out)))
        return self._blocks(out)

    def wrap_greens(self, slice_, direction):
        self._c(lib().dqmc_wrap_greens(self._h, slice_, direction))

    # ---- analysis / measurement sums
    def analysis(self, walker=0):
        st = _lib.Stats()
        self._c(lib().dqmc_get_stats(self._h, walker, C.byref(st)))
        return DQMCAnalysis(st)

    def analysis_sum(self):
        """(prop_local, acc_local) summed over the walkers of this handle"""
        tot = [0, 0]
        for w in range(self.n_walkers):
            a = self.analysis(w)
            tot[0] += a.prop_local
            tot[1] += a.acc_local
        return tuple(tot)

    def accumulate_greens(self):
        self._c(lib().dqmc_accumulate_greens(self._h))

    # ---- measurement reduction over ranks (inside the library: RCCL all-reduce on the engine's stream)
    def reduce(self, comm=None):
        """dqmc_reduce: every accumulator and the DQMCAnalysis counters summed (max / min for the magnitude
        statistics) over all ranks of `comm` (a sharding.Communicator, or None for this handle alone)"""
        self._c(lib().dqmc_reduce(self._h, comm.handle if comm is not None else None))

    def reduced(self, which="greens"):
        """dqmc_get_reduced: the global sums of the last reduction (the handle's own accumulators keep the local sums);
        `which` = greens | correlations | pairing | susceptibilities, layouts as the local getters"""
        idx = {"greens": 0, "correlations": 1, "pairing": 2, "susceptibilities": 3}[which]
        n = C.c_size_t()
        size_fn = (lib().dqmc_accumulator_size, lib().dqmc_correlations_size, lib().dqmc_pairing_size,
                   lib().dqmc_susceptibilities_size)[idx]
        self._c(size_fn(self._h, C.byref(n)))
        out = np.zeros(n.value)
        self._c(lib().dqmc_get_reduced(self._h, idx, dptr(out)))
        return out

    def reduced_analysis(self):
        st = _lib.Stats()
        self._c(lib().dqmc_get_reduced_stats(self._h, C.byref(st)))
        return DQMCAnalysis(st)

    def reduce_size(self):
        n = C.c_size_t()
        self._c(lib().dqmc_reduce_size(self._h, C.byref(n)))
        return n.value

    def reduce_export(self):
        """packed [sums | 2 maxima | 2 minima] for a host-side collective"""
        out = np.zeros(self.reduce_size())
        self._c(lib().dqmc_reduce_export(self._h, dptr(out)))
        return out

    def reduce_import(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        self._c(lib().dqmc_reduce_import(self._h, dptr(buf)))

    def reduce_host(self, dist):
        """the same reduction through a torch.distributed process group on host buffers (gloo): what a host with
        its own collective (MPI from Julia) does around dqmc_reduce_export / dqmc_reduce_import"""
        import torch
        buf = torch.from_numpy(self.reduce_export())
        n = buf.numel() - 4
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(buf[:n], op=dist.ReduceOp.SUM)
            dist.all_reduce(buf[n:n + 2], op=dist.ReduceOp.MAX)
            dist.all_reduce(buf[n + 2:], op=dist.ReduceOp.MIN)
        self.reduce_import(buf.numpy())

    def reset_accumulators(self):
        self._c(lib().dqmc_reset_accumulators(self._h))

    def accumulator_size(self):
        n = C.c_size_t()
        self._c(lib().dqmc_accumulator_size(self._h, C.byref(n)))
        return n.value

    def accumulators(self):
        out = np.zeros(self.accumulator_size())
        self._c(lib().dqmc_get_accumulators(self._h, dptr(out)))
        return out

    def export_accumulators(self, device_ptr):
        self._c(lib().dqmc_export_accumulators(self._h, C.c_void_p(device_ptr)))

    def unpack_accumulators(self, acc):
        """-> dict(G_mean, G2_mean, occupation_mean, count) per block"""
        n, B = self.N, self.nb
        cnt = acc[-1]
        G = [acc[b * n * n:(b + 1) * n * n].reshape((n, n), order="F") / cnt for b in range(B)]
        G2 = [acc[(B + b) * n * n:(B + b + 1) * n * n].reshape((n, n), order="F") / cnt for b in range(B)]
        occ = [acc[2 * B * n * n + b * n:2 * B * n * n + (b + 1) * n] / cnt for b in range(B)]
        return dict(G=G, G2=G2, occupation=occ, count=cnt)

    # ---- equal-time correlations (charge_density_correlation, spin_density_correlation, magnetization)
    def set_pair_directions(self, iterator):
        """hand the EachSitePairByDistance table of the lattice to the device"""
        tab = np.asfortranarray(iterator.dir_of.astype(np.int32))  # [src, trg] -> dir_of[src + n*trg]
        self._ndirs = iterator.ndirections()
        self._c(lib().dqmc_set_pair_directions(self._h, tab.ctypes.data_as(C.POINTER(C.c_int32)), self._ndirs))

    def accumulate_correlations(self):
        self._c(lib().dqmc_accumulate_correlations(self._h))

    def correlations_raw(self):
        """the raw sums [cdc][sdc_x][sdc_y][sdc_z][mx][my][mz][count] (layout of include/dqmc_hip.h)"""
        n = C.c_size_t()
        self._c(lib().dqmc_correlations_size(self._h, C.byref(n)))
        out = np.zeros(n.value)
        self._c(lib().dqmc_get_correlations(self._h, dptr(out)))
        return out

    def correlations(self):
        """-> dict of means: CDC, SDCx, SDCy, SDCz per direction; Mx, My, Mz per site; count"""
        n = C.c_size_t()
        self._c(lib().dqmc_correlations_size(self._h, C.byref(n)))
        out = np.zeros(n.value)
        self._c(lib().dqmc_get_correlations(self._h, dptr(out)))
        nd, N, cnt = self._ndirs, self.N, out[-1]
        names = ["CDC", "SDCx", "SDCy", "SDCz"]
        res = {k: out[i * nd:(i + 1) * nd] / cnt for i, k in enumerate(names)}
        for i, k in enumerate(["Mx", "My", "Mz"]):
            res[k] = out[4 * nd + i * N:4 * nd + (i + 1) * N] / cnt
        res["count"] = cnt
        return res

    # ---- pairing_correlation (measurements.jl:199-214) over EachLocalQuadByDistance{K}
    def set_local_targets(self, iterator):
        """hand the (dir, trg) lists of EachLocalQuadByDistance{K} to the device; the pair
        directions of the same lattice are set with it"""
        self.set_pair_directions(iterator.pairs_by_dir)
        tab = np.asfortranarray(iterator.trg_of.astype(np.int32))  # [src, k] -> trg_of[src + n*k]
        self._K = iterator.K
        self._c(lib().dqmc_set_local_targets(self._h, tab.ctypes.data_as(C.POINTER(C.c_int32)), self._K))

    def accumulate_pairing(self):
        self._c(lib().dqmc_accumulate_pairing(self._h))

    def pairing(self):
        """-> (mean of output[dir12, dir1, dir2] as pushed by finish!, sample count)"""
        n = C.c_size_t()
        self._c(lib().dqmc_pairing_size(self._h, C.byref(n)))
        out = np.zeros(n.value)
        self._c(lib().dqmc_get_pairing(self._h, dptr(out)))
        cnt = out[-1]
        return out[:-1].reshape((self._ndirs, self._K, self._K), order="F") / cnt, cnt

    # ---- unequal-time Green's functions (src/flavors/DQMC/unequal_time_stack.jl)
    def ut_build_stack(self):
        """build_stack(mc, mc.ut_stack)"""
        self._c(lib().dqmc_ut_build_stack(self._h))

    def ut_stack(self, which, idx, walker=0):
        """(U, D, T) blocks of slot idx (0-based) of the forward / backward / inverse stack"""
        sel = {"forward": 0, "backward": 1, "inverse": 2}[which]
        U = np.zeros(self.nb * self.N * self.N); T = np.zeros_like(U); D = np.zeros(self.nb * self.N)
        self._c(lib().dqmc_ut_get_stack(self._h, walker, sel, idx, dptr(U), dptr(D), dptr(T)))
        return self._blocks(U), [D[b * self.N:(b + 1) * self.N] for b in range(self.nb)], self._blocks(T)

    def _ut_result(self, which, walker):
        out = np.zeros(self.nb * self.N * self.N)
        self._c(lib().dqmc_ut_get(self._h, walker, which, dptr(out)))
        return self._blocks(out)

    def calculate_greens_kl(self, slice1, slice2, walker=0):
        """calculate_greens(mc, slice1, slice2): the effective G(slice1 <- slice2)"""
        self._c(lib().dqmc_ut_greens(self._h, slice1, slice2, 1))
        return self._ut_result(0, walker)

    def greens_kl(self, slice1, slice2, walker=None):
        """greens(mc, k, l) = <c_i(k dtau) c_j^dagger(l dtau)>; walker=None returns every walker"""
        self._c(lib().dqmc_ut_greens(self._h, slice1, slice2, 0))
        if walker is None:
            return [self._ut_result(0, w) for w in range(self.n_walkers)]
        return self._ut_result(0, walker)

    def greens_iterator(self, l=0, recalculate=None, walker=0):
        """GreensIterator(mc, :, l, recalculate): yields G(k <- l) for k = l..slices"""
        recalculate = 4 * self.p.safe_mult if recalculate is None else recalculate
        self._c(lib().dqmc_greens_iterator_begin(self._h, l, recalculate))
        yield self._ut_result(0, walker)
        k = C.c_int32()
        while True:
            self._c(lib().dqmc_greens_iterator_next(self._h, C.byref(k)))
            if k.value < 0:
                return
            yield self._ut_result(0, walker)

    def combined_greens_iterator(self, recalculate=None, walker=0):
        """CombinedGreensIterator(mc, recalculate): yields (G0l, Gl0, Gll) for l = 1..slices"""
        recalculate = 4 * self.p.safe_mult if recalculate is None else recalculate
        self._c(lib().dqmc_combined_iterator_begin(self._h, recalculate))
        l = C.c_int32()
        while True:
            self._c(lib().dqmc_combined_iterator_next(self._h, C.byref(l)))
            if l.value < 0:
                return
            yield tuple(self._ut_result(i, walker) for i in range(3))

    def accumulate_susceptibilities(self, recalculate=None):
        """charge_density_/spin_density_/pairing_susceptibility: one pass of the CombinedGreensIterator
        with the packed kernels summed on the device"""
        recalculate = 4 * self.p.safe_mult if recalculate is None else recalculate
        self._c(lib().dqmc_accumulate_susceptibilities(self._h, recalculate))

    def susceptibilities(self):
        """-> dict of means: CDS, SDSx, SDSy, SDSz per direction, PS[dir12, dir1, dir2] if local targets
        are set, count"""
        n = C.c_size_t()
        self._c(lib().dqmc_susceptibilities_size(self._h, C.byref(n)))
        out = np.zeros(n.value)
        self._c(lib().dqmc_get_susceptibilities(self._h, dptr(out)))
        nd, cnt = self._ndirs, out[-1]
        res = {k: out[i * nd:(i + 1) * nd] / cnt for i, k in enumerate(["CDS", "SDSx", "SDSy", "SDSz"])}
        K = getattr(self, "_K", 0)
        if K:
            res["PS"] = out[4 * nd:4 * nd + nd * K * K].reshape((nd, K, K), order="F") / cnt
        res["count"] = cnt
        return res

    # ---- instrumentation
    def qr_fallbacks(self):
        """cooperative-QR launches that timed out and were redone by the single-workgroup kernel"""
        n = C.c_int64()
        self._c(lib().dqmc_qr_fallbacks(self._h, C.byref(n)))
        return n.value

    def udt_one_launch_sites(self):
        """bit mask of the udt_AVX_pivot! call sites served by the one-launch pre-pivoted factorisation (0: none)"""
        w = C.c_int32(0)
        self._c(lib().dqmc_udt_one_launch_sites(self._h, C.byref(w)))
        return int(w.value)

    def device_errors(self):
        """device error word (0 unless a bounded wait inside a kernel ran out)"""
        w = C.c_int32(0)
        self._c(lib().dqmc_device_errors(self._h, C.byref(w)))
        return int(w.value)

    def timing_enable(self, on=True):
        self._c(lib().dqmc_timing_enable(self._h, int(on)))

    def timing(self):
        ms = np.zeros(len(_lib.K_FAMILIES))
        n = np.zeros(len(_lib.K_FAMILIES), dtype=np.int64)
        self._c(lib().dqmc_timing_get(self._h, dptr(ms), i64ptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_lib.K_FAMILIES)}


# ---------------------------------------------------------------------------
# batched linalg primitives (src/linalg), host arrays in/out
def _batch(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        a = a[None]
    return a


def _pack(mats):
    return np.ascontiguousarray(np.concatenate([m.reshape(-1, order="F") for m in mats]))


def _unpack(flat, batch, n):
    return np.stack([flat[i * n * n:(i + 1) * n * n].reshape((n, n), order="F") for i in range(batch)])


def vmul(A, B, transa=False, transb=False, device_id=0):
    """vmul!(C, A, B) and its adjoint/transpose variants (general.jl:7-56)"""
    A, B = _batch(A), _batch(B)
    batch, n = A.shape[0], A.shape[1]
    a, b = _pack(A), _pack(B)
    c = np.zeros_like(a)
    check(lib().dqmc_vmul(device_id, n, batch, int(transa), int(transb), dptr(a), dptr(b), dptr(c)))
    return _unpack(c, batch, n)


def udt_AVX_pivot(X, apply_pivot=True, device_id=0):
    """udt_AVX_pivot!(U, D, T, pivot, temp, Val(apply)) (UDT.jl:192-306) -> U, D, T, pivot(1-based)"""
    X = _batch(X)
    batch, n = X.shape[0], X.shape[1]
    t = _pack(X)
    u = np.zeros_like(t)
    d = np.zeros(batch * n)
    piv = np.zeros(batch * n, dtype=np.int64)
    check(lib().dqmc_udt_pivot(device_id, n, batch, dptr(u), dptr(d), dptr(t), i64ptr(piv), int(apply_pivot)))
    return _unpack(u, batch, n), d.reshape(batch, n), _unpack(t, batch, n), piv.reshape(batch, n)


def rdivp(A, T, pivot, device_id=0):
    """rdivp!(A, T, O, pivot) (general.jl:138-166)"""
    A, T = _batch(A), _batch(T)
    batch, n = A.shape[0], A.shape[1]
    a, t = _pack(A), _pack(T)
    piv = np.ascontiguousarray(np.asarray(pivot, dtype=np.int64).reshape(-1))
    check(lib().dqmc_rdivp(device_id, n, batch, dptr(a), dptr(t), i64ptr(piv)))
    return _unpack(a, batch, n)


def calculate_greens_AVX(Ul, Dl, Tl, Ur, Dr, Tr, device_id=0):
    """calculate_greens_AVX! (stack.jl:337-393)"""
    Ul, Tl, Ur, Tr = _batch(Ul), _batch(Tl), _batch(Ur), _batch(Tr)
    batch, n = Ul.shape[0], Ul.shape[1]
    dl = np.ascontiguousarray(np.asarray(Dl, dtype=np.float64).reshape(-1))
    dr = np.ascontiguousarray(np.asarray(Dr, dtype=np.float64).reshape(-1))
    g = np.zeros(batch * n * n)
    check(lib().dqmc_calculate_greens(device_id, n, batch, dptr(_pack(Ul)), dptr(dl), dptr(_pack(Tl)),
                                      dptr(_pack(Ur)), dptr(dr), dptr(_pack(Tr)), dptr(g)))
    return _unpack(g, batch, n)


def mfma_f64_peak(iters=20000, device_id=0):
    t = C.c_double()
    check(lib().dqmc_mfma_f64_peak(device_id, iters, C.byref(t)))
    return t.value


def device_count():
    return lib().dqmc_device_count()
