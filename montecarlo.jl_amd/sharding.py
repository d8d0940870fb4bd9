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
