"""Reference-side helpers (build container only): construct a reference env for a scenario spec
and dump its state in the id-free canonical form that the oracle and the HIP path are compared in.
"""
import zlib

import numpy as np

import refload


def make_ref_env(spec, recipe, seed=1337):
    m = refload.load()
    from marlgrid.agents import GridAgentInterface
    import marlgrid.envs as E
    cls_name, kwargs = recipe
    def view(a, key):
        return a.get("view", spec)[key]
    agents = [GridAgentInterface(color=a["color"], view_size=view(a, "view_size"),
                                 view_tile_size=view(a, "tile_size"), view_offset=view(a, "view_offset"),
                                 see_through_walls=view(a, "see_through_walls"), spawn_delay=a.get("spawn_delay", 0),
                                 hide_item_types=list(a.get("hide_item_types", [])),
                                 prestige_beta=a.get("prestige_beta", 0.95), prestige_scale=a.get("prestige_scale", 2),
                                 **(dict(observation_style="rich", **a["rich"]) if "rich" in a else {}))
              for a in spec["agents"]]
    kw = dict(kwargs)
    kw.setdefault("max_steps", spec["max_steps"])
    kw["seed"] = seed
    if cls_name == "RegionTestEnv":
        return _region_env_class()(agents=agents, **kw)
    if cls_name == "SpawnRectTestEnv":
        return _spawn_rect_env_class()(agents=agents, **kw)
    if cls_name == "RejectTestEnv":
        return _reject_env_class()(agents=agents, **kw)
    if cls_name == "LateStaticTestEnv":
        return _late_static_env_class()(agents=agents, **kw)
    if cls_name == "KindsTestEnv":
        return _kinds_env_class()(agents=agents, **kw)
    if cls_name == "GroupsTestEnv":
        return _groups_env_class()(agents=agents, **kw)
    return getattr(E, cls_name)(agents=agents, **kw)


def _kinds_env_class():
    """A test-only scenario ON TOP OF the reference's classes: a hundred objects of 115 kinds (scenarios.kinds_list) put on a
    lattice of odd cells, six random walls after them — the registry beyond 64 entries (base.py:19-64: keys are uint8)."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid import objects as RO
    import scenarios

    class KindsTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            for i, (cls, color, kw) in enumerate(scenarios.kinds_list()):
                self.put_obj(getattr(RO, cls)(color=color, **kw), 1 + 2 * (i % 11), 1 + 2 * (i // 11))
            for _ in range(6):
                self.place_obj(RO.Wall(), max_tries=100)
    return KindsTestEnv


def _groups_env_class():
    """A test-only scenario ON TOP OF the reference's classes: sixty random placements of alternating kinds (upstream's
    `_gen_grid` is free Python of any length, base.py:690-708)."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid.objects import Box, Wall

    class GroupsTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            for i in range(60):
                self.place_obj(Wall() if i % 2 == 0 else Box(("red", "blue", "green")[(i // 2) % 3]), max_tries=100)
    return GroupsTestEnv


def _spawn_rect_env_class():
    """A test-only scenario ON TOP OF the reference's classes whose `_gen_grid` does not overwrite
    `agent_spawn_kwargs` (every shipped scenario sets it to {}), so the constructor's kwargs reach
    place_obj(agent, **agent_spawn_kwargs) at reset, late spawn and respawn (base.py:411, 505, 643)."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid.objects import Goal, Wall

    class SpawnRectTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def __init__(self, *a, goal_at=None, **kw):
            self._goal_at = goal_at            # (default: the far corner; `goal_at` puts it where the agents spawn)
            super().__init__(*a, **kw)

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.put_obj(Goal(color="green", reward=1), *(self._goal_at or (2, height - 2)))
            for _ in range(4):
                self.place_obj(Wall(), max_tries=100)
    return SpawnRectTestEnv


def _reject_env_class():
    """A test-only scenario ON TOP OF the reference's classes exercising place_obj(reject_fn=) (base.py:690-708):
    clutter on odd-parity cells only, a locked Door off the diagonal of a 4x4 corner; `agent_spawn_kwargs` (left
    alone by `_gen_grid`) carries a reject_fn too.  The callbacks are the texts in tests/scenarios.py."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid.objects import Goal, Wall, Door
    import scenarios

    class RejectTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)
            for _ in range(5):
                self.place_obj(Wall(), reject_fn=eval(scenarios.REJECT_CLUTTER), max_tries=200)
            self.place_obj(Door(color="yellow", state=3), top=(1, 1), size=(4, 4), reject_fn=eval(scenarios.REJECT_DOOR),
                           max_tries=100)
    return RejectTestEnv


def _late_static_env_class():
    """A test-only scenario ON TOP OF the reference's classes whose `_gen_grid` makes static edits AFTER random
    placements (upstream's `_gen_grid` is free Python: put_obj / wall helpers replace whatever a placement put there,
    base.py:655-662, 160-176) — and places again after them."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid.objects import Goal, Wall

    class LateStaticTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            for _ in range(6):
                self.place_obj(Wall(), max_tries=100)                       # random clutter FIRST ...
            self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)   # ... then the fixed goal (replaces clutter there)
            self.grid.horz_wall(2, height // 2, width - 4)                  # a wall segment over whatever was placed
            self.put_obj(None, 3, height // 2)                              # with a gap (put None)
            self.place_obj(Goal(color="green", reward=1), top=(1, 1), size=(3, 3), max_tries=100)   # and a placement after
            self.grid.wall_rect(width - 4, 1, 3, 3)                         # a 3 x 3 ring in the corner, last
    return LateStaticTestEnv


def _region_env_class():
    """A test-only scenario ON TOP OF the reference's classes exercising place_obj(top=, size=)
    (base.py:690-708): a room split by a wall, a locked Door sampled in the left half, clutter in the right."""
    from marlgrid.base import MultiGridEnv, MultiGrid
    from marlgrid.objects import Goal, Wall, Door

    class RegionTestEnv(MultiGridEnv):
        mission = ""
        metadata = {}

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.grid.vert_wall(width // 2, 0, height - 3)
            self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)
            self.place_obj(Door(color="yellow", state=3), top=(0, 0), size=(width // 2, height), max_tries=100)
            for _ in range(3):
                self.place_obj(Wall(), top=(width // 2 + 1, 2), size=(width, height - 3), max_tries=50)
    return RegionTestEnv


def canonical(env):
    """id-free canonical state of a reference env."""
    W, H, n = env.width, env.height, len(env.agents)
    base_enc = np.zeros((W, H, 3), np.uint8)
    for i in range(W):
        for j in range(H):
            o = env.grid.get(i, j)
            if o is not None and not o.is_agent:
                base_enc[i, j] = o.encode()
    pos = np.full((n, 2), -1, np.int16)
    d = np.zeros(n, np.int8)
    active = np.zeros(n, bool)
    done = np.zeros(n, bool)
    carry = np.zeros((n, 3), np.uint8)
    ordinal = np.full(n, -1, np.int8)
    for k, a in enumerate(env.agents):
        d[k], active[k], done[k] = a.dir, a.active, a.done
        if a.carrying is not None:
            carry[k] = a.carrying.encode()
        if a.pos is not None:
            pos[k] = a.pos
            o = env.grid.get(*a.pos)
            if o is a:
                ordinal[k] = 0
            elif o is None or a not in o.agents:
                ordinal[k] = -1          # put_obj replaced the cell it stood on (base.py:655-662): in no cell any more
            elif o.is_agent:
                ordinal[k] = 1 + o.agents.index(a)
            else:
                ordinal[k] = o.agents.index(a)
    return dict(base_enc=base_enc, pos=pos, dir=d, active=active, done=done, carry_enc=carry,
                ordinal=ordinal, step_count=int(env.step_count))


class OrderSpy(object):
    """Proxy around env.np_random recording each shuffled iteration order (base.py:514-516)."""

    def __init__(self, rng):
        self._rng = rng
        self.last = None

    def shuffle(self, x):
        self._rng.shuffle(x)
        self.last = np.array(x).copy()

    def __getattr__(self, k):
        return getattr(self._rng, k)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.uint8).tobytes()) & 0xFFFFFFFF
