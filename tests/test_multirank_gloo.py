"""The N > 1 path on CPU: two gloo ranks shard the walkers, run them with the CPU oracle
(standing in for the per-GPU engines), build the accumulator layout of include/dqmc_hip.h
and all-reduce it; the result must equal the single-process sum over all walkers, and the
per-walker trajectories must not depend on the sharding."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def walker_accumulator(O, seed, L=2, beta=1.0, sweeps=3):
    mc = O.OracleDQMC(L, "attractive", beta=beta)
    mc.set_conf(O.random_conf(seed, L * L, mc.slices))
    mc.seed(seed)
    mc.prepare()
    n = mc.N
    acc = np.zeros(2 * n * n + n + 1)
    for _ in range(sweeps):
        mc.update_until_measure()
        g = mc.greens()[0].reshape(-1, order="F")
        acc[:n * n] += g
        acc[n * n:2 * n * n] += g * g
        acc[2 * n * n:2 * n * n + n] += 1.0 - np.diag(mc.greens()[0])
        acc[-1] += 1
    return acc, mc.conf()


def _worker(rank, world, port, per_rank, ret):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    mc_amd = g.load_package()
    O = g.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = None
    confs = {}
    for s in mc_amd.walker_seeds(123, rank, world, per_rank):
        a, c = walker_accumulator(O, s)
        acc = a if acc is None else acc + a
        confs[s] = c
    t = torch.from_numpy(acc.copy())
    mc_amd.reduce_accumulators(t, dist)
    ret[rank] = (t.numpy().copy(), confs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_matches_single_process():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    O = g.load_oracle()
    world, per_rank = 2, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, per_rank, ret), nprocs=world, join=True)
    total, confs = None, {}
    for s in range(123, 123 + world * per_rank):
        a, c = walker_accumulator(O, s)
        total = a if total is None else total + a
        confs[s] = c
    for rank in range(world):
        red, cf = ret[rank]
        assert np.allclose(red, total, rtol=1e-13, atol=1e-13)
        assert red[-1] == 3 * world * per_rank
        for s, c in cf.items():
            assert np.array_equal(c, confs[s]), "trajectory of walker seed %d depends on the sharding" % s
