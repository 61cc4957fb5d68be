"""CPU, world_size 2, gloo: the N>1 path of bench.py that is not GPU work — shard arithmetic
(disjoint cover, seeds by global env id) and the max/sum-over-ranks reductions."""
import os
import socket
import sys

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from marlgrid_amd import sharding
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    G = 262144 + 3
    lo, hi = sharding.shard_range(G, rank, world)
    seeds = sharding.shard_seeds(1337, G, rank, world)
    t = sharding.max_over_ranks(1.0 + rank)              # slowest rank wins
    total = sharding.sum_over_ranks(hi - lo)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, seeds[0], seeds[-1], t, total))


def test_two_rank_sharding_and_reduction():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G = 262144 + 3
    (r0, lo0, hi0, s0a, s0b, t0, tot0), (r1, lo1, hi1, s1a, s1b, t1, tot1) = out
    assert lo0 == 0 and hi0 == lo1 and hi1 == G                # disjoint, contiguous cover
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert (s0a, s0b, s1a, s1b) == (1337, 1337 + hi0 - 1, 1337 + lo1, 1337 + G - 1)
    assert t0 == t1 == 2.0                                     # MAX over ranks
    assert tot0 == tot1 == G


def test_shard_range_properties():
    from marlgrid_amd.sharding import shard_range
    for G in (1, 7, 8, 262144, 1048576 + 5):
        for W in (1, 2, 4, 8):
            prev = 0
            for r in range(W):
                lo, hi = shard_range(G, r, W)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == G
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)
