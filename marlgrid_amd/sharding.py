"""Multi-GPU sharding of the env batch: one process per GPU, contiguous env slices, no collective
on the data path (envs are fully independent: own grid, agents and RNG — marlgrid/base.py:371-374).
Only the benchmark's wall-clock is reduced over ranks (MAX)."""


def shard_range(global_batch, rank, world_size):
    """Contiguous slice [lo, hi) of the global env ids owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(base_seed, global_batch, rank, world_size):
    """Per-env seeds of this rank's slice: env with global id g is seeded base_seed + g, whatever the
    sharding — which is what makes trajectories shard-invariant."""
    lo, hi = shard_range(global_batch, rank, world_size)
    return [int(base_seed) + g for g in range(lo, hi)]


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over the default process group (returns it unchanged when
    torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
