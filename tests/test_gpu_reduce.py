"""Measurement reduction behind the C ABI (dqmc_reduce) with the REAL engine in every rank:
 * two handles that shard 2W walkers (first_walker 0 and W) reproduce the trajectories of one 2W-walker handle, and
   their packed reduce buffers, combined as the collective would (sum | max | min), equal the big handle's;
 * the RCCL path (ncclAllReduce inside the library) on a one-rank communicator;
 * two processes on one device (functional only) through the host-mediated export -> gloo all-reduce -> import."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _measure(mc, sweeps):
    for _ in range(sweeps):
        mc.update_until_measure()
        mc.accumulate_greens()
        mc.accumulate_correlations()


def _run(gpu, first, W, model_cls=None, L=4, beta=2.0, sweeps=3):
    model = (model_cls or gpu.HubbardModelRepulsive)(L, 2)
    mc = gpu.DQMC(model, beta=beta, n_walkers=W, seed=123, first_walker=first)
    mc.set_pair_directions(gpu.EachSitePairByDistance(model.l))
    mc.prepare()
    _measure(mc, sweeps)
    return mc


def _combine(bufs):
    n = len(bufs[0]) - 4
    out = np.sum([b[:n] for b in bufs], axis=0)
    mx = np.max([b[n:n + 2] for b in bufs], axis=0)
    mn = np.min([b[n + 2:] for b in bufs], axis=0)
    return np.concatenate([out, mx, mn])


def test_sharded_handles_equal_one_big_handle(gpu):
    W = 3
    big = _run(gpu, 0, 2 * W)
    a, b = _run(gpu, 0, W), _run(gpu, W, W)
    for w in range(W):  # trajectories do not depend on the sharding (seeds keyed by global walker id)
        assert np.array_equal(a.conf(w), big.conf(w))
        assert np.array_equal(b.conf(w), big.conf(W + w))
    ref = big.reduce_export()
    comb = _combine([a.reduce_export(), b.reduce_export()])
    assert np.allclose(comb, ref, rtol=1e-12, atol=1e-12)
    # import the combined buffer into one shard: dqmc_get_reduced / dqmc_get_reduced_stats give the global sums, the
    # shard's own accumulators keep its local sums
    local = a.accumulators().copy()
    a.reduce_import(comb)
    assert np.allclose(a.reduced("greens"), big.accumulators(), rtol=1e-12, atol=1e-12)
    assert np.allclose(a.reduced("correlations"), big.correlations_raw(), rtol=1e-12, atol=1e-12)
    assert np.array_equal(a.accumulators(), local)
    ra = a.reduced_analysis()
    tot = big.analysis_sum()
    assert (ra.prop_local, ra.acc_local) == tot
    # the reduction is repeated every measure_rate sweeps (DQMC.jl:429-436): a SECOND reduction after more samples
    # must again equal the big handle (writing the global sums back into the shards would count the first three
    # samples twice here)
    for m in (big, a, b):
        _measure(m, 2)
    comb2 = _combine([a.reduce_export(), b.reduce_export()])
    assert np.allclose(comb2, big.reduce_export(), rtol=1e-12, atol=1e-12)
    for m in (a, b):
        m.reduce_import(comb2)
        assert np.allclose(m.reduced("greens"), big.accumulators(), rtol=1e-12, atol=1e-12)
        assert np.allclose(m.reduced("correlations"), big.correlations_raw(), rtol=1e-12, atol=1e-12)
        assert m.reduced("greens")[-1] == 5 * 2 * W
        ra = m.reduced_analysis()
        assert (ra.prop_local, ra.acc_local) == big.analysis_sum()
    for m in (big, a, b):
        m.close()


def test_rccl_single_rank_reduce(gpu):
    """ncclAllReduce inside the library (one-rank communicator): values unchanged, counters = local sums"""
    L = gpu.lib()
    idbuf = (C.c_ubyte * 128)()
    gpu._lib.check(L.dqmc_comm_unique_id(C.cast(idbuf, C.c_void_p)))
    comm = C.c_void_p()
    gpu._lib.check(L.dqmc_comm_init(C.cast(idbuf, C.c_void_p), 1, 0, 0, C.byref(comm)))
    mc = _run(gpu, 0, 2, gpu.HubbardModelAttractive)
    before = mc.accumulators().copy()

    class Cm:
        handle = comm
    mc.reduce(Cm)
    assert np.array_equal(mc.accumulators(), before)
    assert np.array_equal(mc.reduced("greens"), before)
    ra = mc.reduced_analysis()
    assert (ra.prop_local, ra.acc_local) == mc.analysis_sum()
    mc.reduce(Cm)  # again, nothing accumulated in between: same sums (not doubled)
    assert np.array_equal(mc.reduced("greens"), before)
    mc.reduce(None)
    assert np.array_equal(mc.accumulators(), before)
    assert np.array_equal(mc.reduced("correlations"), mc.correlations_raw())
    mc.close()
    gpu._lib.check(L.dqmc_comm_destroy(comm))


def _worker(rank, world, port, W, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as g
    gpu = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = gpu.walker_block(rank, world, world * W)
    mc = _run(gpu, lo, hi - lo)
    mc.reduce_host(dist)
    _measure(mc, 1)
    mc.reduce_host(dist)  # second reduction of the same run (3 + 1 samples per walker)
    ra = mc.reduced_analysis()
    ret[rank] = (mc.reduced("greens"), mc.reduced("correlations"), (ra.prop_local, ra.acc_local),
                 [mc.conf(w) for w in range(hi - lo)])
    mc.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_device(gpu):
    import torch.multiprocessing as mp
    world, W = 2, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, W, ret), nprocs=world, join=True)
    big = _run(gpu, 0, world * W, sweeps=4)
    for rank in range(world):
        acc, corr, cnt, confs = ret[rank]
        assert np.allclose(acc, big.accumulators(), rtol=1e-12, atol=1e-12)
        assert np.allclose(corr, big.correlations_raw(), rtol=1e-12, atol=1e-12)
        assert cnt == big.analysis_sum()
        for w, c in enumerate(confs):
            assert np.array_equal(c, big.conf(rank * W + w))
    big.close()
