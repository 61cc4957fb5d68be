"""Build the PRODUCT env (marlgrid_amd, HIP path) for a named test scenario."""
import numpy as np

import canon


def build(name, **kw):
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd import envs as E
    import scenarios
    if name in E._registry:
        return E.make(name, **kw)
    spec = scenarios.registered(name)
    cls_name, kwargs = scenarios.ref_recipe(name)
    def view(a, key):
        return a.get("view", spec)[key]
    agents = [GridAgentInterface(color=a["color"], view_size=view(a, "view_size"), view_tile_size=view(a, "tile_size"),
                                 view_offset=view(a, "view_offset"), see_through_walls=view(a, "see_through_walls"),
                                 spawn_delay=a.get("spawn_delay", 0),
                                 hide_item_types=list(a.get("hide_item_types", [])),
                                 prestige_beta=a.get("prestige_beta", 0.95), prestige_scale=a.get("prestige_scale", 2),
                                 **(dict(observation_style="rich", **a["rich"]) if "rich" in a else {}))
              for a in spec["agents"]]
    if cls_name == "RegionTestEnv":
        return _region_env_class()(agents=agents, **{**kwargs, **kw})
    if cls_name == "SpawnRectTestEnv":
        return _spawn_rect_env_class()(agents=agents, **{**kwargs, **kw})
    if cls_name == "RejectTestEnv":
        return _reject_env_class()(agents=agents, **{**kwargs, **kw})
    if cls_name == "LateStaticTestEnv":
        return _late_static_env_class()(agents=agents, **{**kwargs, **kw})
    if cls_name == "KindsTestEnv":
        return _kinds_env_class()(agents=agents, **{**kwargs, **kw})
    if cls_name == "GroupsTestEnv":
        return _groups_env_class()(agents=agents, **{**kwargs, **kw})
    return getattr(E, cls_name)(agents=agents, **{**kwargs, **kw})


def _kinds_env_class():
    """the product-side twin of tests/golden/refstate.py:_kinds_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd import objects as RO
    import scenarios

    class KindsTestEnv(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            for i, (cls, color, kw) in enumerate(scenarios.kinds_list()):
                self.put_obj(getattr(RO, cls)(color=color, **kw), 1 + 2 * (i % 11), 1 + 2 * (i // 11))
            for _ in range(6):
                self.place_obj(RO.Wall(), max_tries=100)
    return KindsTestEnv


def _groups_env_class():
    """the product-side twin of tests/golden/refstate.py:_groups_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Box, Wall

    class GroupsTestEnv(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            for i in range(60):
                self.place_obj(Wall() if i % 2 == 0 else Box(("red", "blue", "green")[(i // 2) % 3]), max_tries=100)
    return GroupsTestEnv


def _region_env_class():
    """the product-side twin of tests/golden/refstate.py:_region_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Goal, Wall, Door

    class RegionTestEnv(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.grid.vert_wall(width // 2, 0, height - 3)
            self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)
            self.place_obj(Door(color="yellow", state=3), top=(0, 0), size=(width // 2, height), max_tries=100)
            for _ in range(3):
                self.place_obj(Wall(), top=(width // 2 + 1, 2), size=(width, height - 3), max_tries=50)
    return RegionTestEnv


def _late_static_env_class():
    """the product-side twin of tests/golden/refstate.py:_late_static_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Goal, Wall

    class LateStaticTestEnv(MultiGridEnv):
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


def _reject_env_class():
    """the product-side twin of tests/golden/refstate.py:_reject_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Goal, Wall, Door
    import scenarios

    class RejectTestEnv(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)
            for _ in range(5):
                self.place_obj(Wall(), reject_fn=eval(scenarios.REJECT_CLUTTER), max_tries=200)
            self.place_obj(Door(color="yellow", state=3), top=(1, 1), size=(4, 4), reject_fn=eval(scenarios.REJECT_DOOR),
                           max_tries=100)
    return RejectTestEnv


def _spawn_rect_env_class():
    """the product-side twin of tests/golden/refstate.py:_spawn_rect_env_class (same _gen_grid text)"""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Goal, Wall

    class SpawnRectTestEnv(MultiGridEnv):
        def __init__(self, *a, goal_at=None, **kw):
            self._goal_at = goal_at
            super().__init__(*a, **kw)

        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.put_obj(Goal(color="green", reward=1), *(self._goal_at or (2, height - 2)))
            for _ in range(4):
                self.place_obj(Wall(), max_tries=100)
    return SpawnRectTestEnv


def canonical(env, b=None):
    """canonical id-free state of every env (list) or env b (dict)"""
    out = canonical_arrays(env.scenario_spec(), env.grid.grid.cpu().numpy(), env.agent_state.cpu().numpy(),
                           env.step_count.cpu().numpy())
    return out if b is None else out[b]


def canonical_arrays(spec, base, rec, sc):
    """the same from plain arrays: base (B, W, H) object ids, rec (B, n) packed agent records, sc (B,)"""
    import _native_consts as K
    rec = np.asarray(rec).astype(np.uint64)
    B, n = rec.shape
    by = lambda i: ((rec >> np.uint64(8 * i)) & np.uint64(0xFF)).astype(np.int64)
    x, y, d, fl, ca, rk = by(K.AG_X), by(K.AG_Y), by(K.AG_DIR), by(K.AG_FLAGS), by(K.AG_CARRY), by(K.AG_RANK)
    placed = (fl & K.AF_PLACED) != 0
    out = []
    for bb in range(B):
        there = placed[bb] | ((fl[bb] & K.AF_EVICTED) != 0)     # (an evicted agent keeps its position; ordinal -1)
        pos = np.stack([np.where(there, x[bb], -1), np.where(there, y[bb], -1)], axis=1)
        ordinal = np.full(n, -1)
        for k in range(n):
            if placed[bb, k]:
                same = placed[bb] & (x[bb] == x[bb, k]) & (y[bb] == y[bb, k])
                ordinal[k] = int((same & (rk[bb] < rk[bb, k])).sum())
        out.append(canon.from_ids(spec, base[bb], pos, d[bb], (fl[bb] & K.AF_ACTIVE) != 0,
                                  (fl[bb] & K.AF_DONE) != 0, ca[bb], ordinal, sc[bb]))
    return out
