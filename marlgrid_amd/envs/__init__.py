"""Scenario classes + the registered ids (marlgrid/envs/__init__.py).

`gym` is not a dependency here: `register_marl_env` fills a module-level registry and
`make(id, batch_size=..., device=...)` plays the role of `gym.make`.
`DoorKeyEnv` is not provided: upstream's cannot be constructed (`self._rand_int` is undefined,
doorkey.py:26,34).
"""
import inspect
import random
import sys

from ..agents import GridAgentInterface
from ..base import MultiGridEnv
from .scenarios import ClutteredGoalCycleEnv, ClutteredMultiGrid, EmptyMultiGrid, VisibilityTestEnv

this_module = sys.modules[__name__]
registered_envs = []
_registry = {}


def register_marl_env(env_name, env_class, n_agents, grid_size, view_size, view_tile_size=8, view_offset=0,
                      agent_color=None, env_kwargs={}):
    colors = ["red", "blue", "purple", "orange", "olive", "pink"]
    assert n_agents <= len(colors)

    def build(**extra):
        agents = [GridAgentInterface(color=c if agent_color is None else agent_color, view_size=view_size,
                                     view_tile_size=8,        # upstream passes the literal 8 (__init__.py:42)
                                     view_offset=view_offset)
                  for c in colors[:n_agents]]
        return env_class(agents=agents, grid_size=grid_size, **{**env_kwargs, **extra})

    build.__name__ = "env_%d" % len(registered_envs)
    setattr(this_module, build.__name__, build)
    registered_envs.append(env_name)
    _registry[env_name] = build


def make(env_name, **kwargs):
    """gym.make stand-in; extra kwargs (batch_size, device, seed, seeds, auto_reset, strict, ...) go to
    the env constructor."""
    if env_name not in _registry:
        raise KeyError("unknown env id %r; registered: %s" % (env_name, ", ".join(registered_envs)))
    return _registry[env_name](**kwargs)


def env_from_config(env_config, randomize_seed=True):
    possible_envs = {k: v for k, v in globals().items() if inspect.isclass(v) and issubclass(v, MultiGridEnv)}
    env_class = possible_envs[env_config["env_class"]]
    env_kwargs = {k: v for k, v in env_config.items() if k != "env_class"}
    if randomize_seed:
        env_kwargs["seed"] = env_kwargs.get("seed", 0) + random.randint(0, 1337 * 1337)
    return env_class(**env_kwargs)


# the ids upstream registers (marlgrid/envs/__init__.py:70-121), as data.  Note that upstream's
# "1AgentCluttered15x15" really is an 11x11 grid with view 5.
_SHIPPED = [
    ("MarlGrid-1AgentCluttered15x15-v0", ClutteredMultiGrid, dict(n_agents=1, grid_size=11, view_size=5,
                                                                  env_kwargs={"n_clutter": 30})),
    ("MarlGrid-3AgentCluttered11x11-v0", ClutteredMultiGrid, dict(n_agents=3, grid_size=11, view_size=7,
                                                                  env_kwargs={"clutter_density": 0.15})),
    ("MarlGrid-3AgentCluttered15x15-v0", ClutteredMultiGrid, dict(n_agents=3, grid_size=15, view_size=7,
                                                                  env_kwargs={"clutter_density": 0.15})),
    ("MarlGrid-2AgentEmpty9x9-v0", EmptyMultiGrid, dict(n_agents=2, grid_size=9, view_size=7)),
    ("MarlGrid-3AgentEmpty9x9-v0", EmptyMultiGrid, dict(n_agents=3, grid_size=9, view_size=7)),
    ("MarlGrid-4AgentEmpty9x9-v0", EmptyMultiGrid, dict(n_agents=4, grid_size=9, view_size=7)),
    ("Goalcycle-demo-solo-v0", ClutteredGoalCycleEnv, dict(n_agents=1, grid_size=13, view_size=7, view_tile_size=5,
                                                            view_offset=1,
                                                            env_kwargs={"clutter_density": 0.1, "n_bonus_tiles": 3})),
]
for _name, _cls, _kw in _SHIPPED:
    register_marl_env(_name, _cls, **_kw)
