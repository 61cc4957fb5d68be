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


class ShardPipeline(object):
    """The batch of ONE GPU as `parts` independent envs, each stepped on its own stream.

    Why: a launch of the step kernel has a store-free head (staging, the env step, the first views: ~27 us during
    which HBM idles) and a ragged tail (waves exit between 80 and 165 us of a 165 us launch).  Two launches that do
    not depend on each other overlap there — the later one's workgroups start on the CUs the earlier one's leave —
    and the envs of a batch ARE independent (base.py:371-374: one RNG per env, nothing shared).  Measured on MI355X
    (profiles/r03/two_halves*.txt): 2 x 16 384 envs on two streams step 6-8 % faster than one env of 32 768,
    2 x 32 768 17 % faster than one of 65 536 (0.289 against 0.347 ms per step: 682 M agent-steps/s on one GPU); the
    same parts on ONE stream are 13 % slower than the one env, and parts that are joined after every step gain
    nothing — the overlap is between step i of one part and step i + 1 of the other, so it is there for callers that
    drive the parts independently (the usual double-buffered sampler: the policy looks at part A's observations
    while part B steps), not for a caller that needs all observations of a step before it issues the next.

    Env g of the global batch keeps its seed (seed + g, marlgrid_amd.sharding.shard_seeds), so trajectories are
    bit-identical to those of the one big env.  Everything a part returns is ordered on ITS stream
    (`pipe.streams[k]`): consume it there, or `pipe.streams[k].synchronize()` / `pipe.synchronize()` first.
    """

    def __init__(self, make_env, batch_size, parts=2, seed=1337, device=None, streams=None):
        """make_env(batch_size=..., seeds=..., device=...) -> MultiGridEnv.  `streams`: one per part (default: new
        ones.  HIP multiplexes streams onto a few hardware queues — GPU_MAX_HW_QUEUES, 4 by default —, and two streams
        that share a queue do not overlap: a process that builds several pipelines should hand the same streams to
        all of them)."""
        import torch
        if parts < 1 or batch_size % parts:
            raise ValueError("batch_size must be a multiple of parts")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch_size, self.parts = int(batch_size), int(parts)
        self.part_size = self.batch_size // self.parts
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(device=self.device) for _ in range(self.parts)]
        if len(self.streams) != self.parts:
            raise ValueError("one stream per part")
        # Part k is BUILT on the stream it will be stepped on: a constructor ends with launches nobody waits for (the
        # first observation; the seeding and reset of a non-strict env), and torch's side streams do not wait for the
        # stream that happened to be current — on its own stream the part's first reset() / step_part() simply queues
        # behind them.
        self.envs = []
        for k in range(self.parts):
            with torch.cuda.stream(self.streams[k]):
                self.envs.append(make_env(batch_size=self.part_size, seeds=shard_seeds(seed, self.batch_size, k, self.parts),
                                          device=self.device))

    # what a training loop asks an env for
    @property
    def num_agents(self):
        return self.envs[0].num_agents

    @property
    def agents(self):
        """part 0's agent interfaces (every part has its own, bound to its own env: agent.pos etc. are per part)"""
        return self.envs[0].agents

    @property
    def action_space(self):
        return self.envs[0].action_space

    @property
    def observation_space(self):
        return self.envs[0].observation_space

    def on(self, k):
        """`with pipe.on(k): ...` — part k's stream as torch's current stream: the policy's kernels for part k and
        everything else that consumes its observations belong here, so that they queue behind part k's step and
        overlap the OTHER part's (the double-buffered sampler:

            obs = pipe.reset()
            while ...:
                for k in range(pipe.parts):
                    with pipe.on(k):
                        act = agents.action_step(obs[k])
                        nxt, rew, done, _ = pipe.step_part(k, act)
                        agents.save_step(obs[k], act, nxt, rew, done)
                        obs[k] = nxt
        )"""
        import torch
        return torch.cuda.stream(self.streams[k])

    def reset_part(self, k, **kw):
        with self.on(k):
            return self.envs[k].reset(**kw)

    def _each(self, fn):
        import torch
        out = []
        for k, env in enumerate(self.envs):
            with torch.cuda.stream(self.streams[k]):
                out.append(fn(k, env))
        return out

    def part(self, k, tensor):
        """rows of part k in a (batch_size, ...) tensor"""
        return tensor[k * self.part_size:(k + 1) * self.part_size]

    def reset(self):
        """per part: its observations (ordered on streams[k])"""
        return self._each(lambda k, env: env.reset())

    def step(self, actions):
        """actions: (batch_size, n_agents), resident and ready (the parts' streams do not wait for the stream that
        produced it), or a list with one tensor per part.  Returns per part (obs, rewards, done, info), each ordered on
        streams[k]."""
        per_part = isinstance(actions, (list, tuple))
        return self._each(lambda k, env: env.step(actions[k] if per_part else self.part(k, actions)))

    def step_part(self, k, actions):
        """one part alone (the double-buffered sampler's call)"""
        import torch
        with torch.cuda.stream(self.streams[k]):
            return self.envs[k].step(actions)

    def close(self):
        self.synchronize()
        self.envs = []

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def check_errors(self):
        """every part's check_errors(), each under its own stream (an env waits for every stream it launched on, so
        this also holds for parts that were stepped through step_part from elsewhere)"""
        self._each(lambda k, env: env.check_errors())
