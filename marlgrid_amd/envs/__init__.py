"""Scenario classes and the ids the reference registers with gym (marlgrid/envs/__init__.py:20-121).

gym is not a dependency: ids live in a module-level table and `make(id, batch_size=..., device=...)`
stands in for `gym.make`.  `DoorKeyEnv` is absent on purpose — upstream's cannot be constructed
(`self._rand_int` is undefined, doorkey.py:26,34).
"""
import functools
import random

from ..agents import GridAgentInterface
from ..base import MultiGridEnv
from .scenarios import ClutteredGoalCycleEnv, ClutteredMultiGrid, EmptyMultiGrid, VisibilityTestEnv

_PALETTE = ("red", "blue", "purple", "orange", "olive", "pink")     # per-slot agent colours of a registered id
_registry = {}            # id -> factory(**constructor kwargs)
registered_envs = []      # ids, in registration order (upstream's list of the same name)


def _construct(env_class, n_agents, geometry, agent_color, fixed, **extra):
    view_size, view_offset = geometry
    team = [GridAgentInterface(color=agent_color or _PALETTE[i], view_size=view_size, view_offset=view_offset,
                               view_tile_size=8)           # upstream hard-codes 8 here whatever it was given (:42)
            for i in range(n_agents)]
    return env_class(agents=team, **dict(fixed, **extra))


def register_marl_env(env_name, env_class, n_agents, grid_size, view_size, view_tile_size=8, view_offset=0,
                      agent_color=None, env_kwargs={}):
    """Same signature as upstream (:20-55).  `view_tile_size` is accepted and — as upstream — ignored."""
    if n_agents > len(_PALETTE):
        raise AssertionError("a registered id has at most %d agents" % len(_PALETTE))
    _registry[env_name] = functools.partial(_construct, env_class, n_agents, (view_size, view_offset), agent_color,
                                            dict(env_kwargs, grid_size=grid_size))
    registered_envs.append(env_name)


def make(env_name, pipeline=None, **kwargs):
    """`gym.make` stand-in; kwargs (batch_size, device, seed, seeds, auto_reset, strict, ...) reach the env.

    pipeline=P (P >= 2): the batch as P independent envs of batch_size / P on P streams — a
    `marlgrid_amd.sharding.ShardPipeline` whose parts a sampler steps in turn (`pipe.step_part(k, actions)` under
    `pipe.on(k)`): the launches of independent shards overlap (+10 % at 32 768 envs, +17 % at 65 536 on one
    MI355X), and env g of the batch keeps its seed `seed + g`, so trajectories are those of the one big env."""
    try:
        factory = _registry[env_name]
    except KeyError:
        raise KeyError("unknown env id %r; registered: %s" % (env_name, ", ".join(registered_envs))) from None
    if pipeline is None or int(pipeline) <= 1:
        return factory(**kwargs)
    from ..sharding import ShardPipeline
    if "seeds" in kwargs:
        raise ValueError("make(pipeline=): per-env seeds come from `seed` + the env's index in the whole batch")
    batch_size, seed, device = kwargs.pop("batch_size"), kwargs.pop("seed", 1337), kwargs.pop("device", None)
    streams = kwargs.pop("streams", None)
    return ShardPipeline(lambda batch_size, seeds, device: factory(batch_size=batch_size, seeds=seeds, device=device, **kwargs),
                         batch_size, parts=int(pipeline), seed=seed, device=device, streams=streams)


def _scenario_classes():
    found, todo = {}, [MultiGridEnv]
    while todo:
        cls = todo.pop()
        found[cls.__name__] = cls
        todo.extend(cls.__subclasses__())
    return found


def env_from_config(env_config, randomize_seed=True):
    """Build the scenario class named by `env_config["env_class"]` from the rest of the dict (:58-67)."""
    spec = dict(env_config)
    env_class = _scenario_classes()[spec.pop("env_class")]
    if randomize_seed:
        spec["seed"] = spec.get("seed", 0) + random.randint(0, 1337 * 1337)
    return env_class(**spec)


# (id, class, agents, grid, view, view_offset, scenario kwargs) — upstream :70-121.  Its
# "1AgentCluttered15x15" really is an 11x11 room seen through a 5x5 view.
for _row in (
        ("MarlGrid-1AgentCluttered15x15-v0", ClutteredMultiGrid, 1, 11, 5, 0, dict(n_clutter=30)),
        ("MarlGrid-3AgentCluttered11x11-v0", ClutteredMultiGrid, 3, 11, 7, 0, dict(clutter_density=0.15)),
        ("MarlGrid-3AgentCluttered15x15-v0", ClutteredMultiGrid, 3, 15, 7, 0, dict(clutter_density=0.15)),
        ("MarlGrid-2AgentEmpty9x9-v0", EmptyMultiGrid, 2, 9, 7, 0, {}),
        ("MarlGrid-3AgentEmpty9x9-v0", EmptyMultiGrid, 3, 9, 7, 0, {}),
        ("MarlGrid-4AgentEmpty9x9-v0", EmptyMultiGrid, 4, 9, 7, 0, {}),
        ("Goalcycle-demo-solo-v0", ClutteredGoalCycleEnv, 1, 13, 7, 1, dict(clutter_density=0.1, n_bonus_tiles=3)),
):
    register_marl_env(_row[0], _row[1], n_agents=_row[2], grid_size=_row[3], view_size=_row[4], view_offset=_row[5],
                      env_kwargs=_row[6])
del _row
