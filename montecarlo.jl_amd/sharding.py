"""Walker sharding over GPUs (one process per GPU) and the measurement reduction.

Markov chains never interact (the reference has exactly one chain per process), so
walkers are dealt to ranks in contiguous blocks and the only collective is the sum of
the measurement accumulators (RCCL all-reduce when the backend is "nccl")."""


def walker_range(rank, world_size, walkers_per_rank):
    """global walker ids owned by `rank` (weak scaling: every rank holds walkers_per_rank)"""
    first = rank * walkers_per_rank
    return first, list(range(first, first + walkers_per_rank))


def walker_seeds(base_seed, rank, world_size, walkers_per_rank):
    """walker w uses seed base_seed + w for its initial HS field and its Metropolis stream,
    whatever the number of ranks (results are independent of the sharding)"""
    _, ids = walker_range(rank, world_size, walkers_per_rank)
    return [base_seed + w for w in ids]


def reduce_accumulators(acc, dist=None):
    """in-place sum of a torch accumulator tensor over all ranks; identity for one rank"""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


def walker_block(rank, world_size, total_walkers):
    """[lo, hi) of the global walker ids owned by `rank` when a FIXED number of walkers is split over the ranks
    (strong scaling, BASELINE config 4): contiguous blocks, the first total % world ranks hold one more"""
    base, rem = divmod(total_walkers, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Communicator:
    """RCCL communicator owned by libdqmc_hip.so (dqmc_comm_init).  torch.distributed is only the out-of-band
    channel that carries rank 0's 128-byte ncclUniqueId to the other ranks, as MPI would for a Julia host."""

    def __init__(self, dist, device_id=None):
        import ctypes as C
        from ._lib import check, lib
        import torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        dev = torch.cuda.current_device() if device_id is None else device_id
        idbuf = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(lib().dqmc_comm_unique_id(C.cast(idbuf, C.c_void_p)))
        obj = [bytes(idbuf)]
        dist.broadcast_object_list(obj, src=0)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(obj[0])
        self.handle = C.c_void_p()
        check(lib().dqmc_comm_init(C.cast(idbuf, C.c_void_p), self.world, self.rank, dev, C.byref(self.handle)))

    def close(self):
        from ._lib import lib
        if self.handle:
            lib().dqmc_comm_destroy(self.handle)
            self.handle = None
