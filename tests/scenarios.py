"""Plain-data scenario specs for the oracle, restating `marlgrid/envs/*.py` + the registered ids
(`marlgrid/envs/__init__.py:70-121`).  Independent of the product package on purpose: the product
emits its own spec (`MultiGridEnv.scenario_spec()`), and a test asserts the two agree.
"""
REG_COLORS = ["red", "blue", "purple", "orange", "olive", "pink"]   # envs/__init__.py:30

WALL = dict(type="Wall", color="worst", state=0)                    # objects.py:47,280
GOAL = dict(type="Goal", color="green", state=0, reward=1)          # cluttered.py:29, empty.py:12


def _base(n_agents, grid_size, view_size, tile_size=8, view_offset=0, colors=None, **kw):
    colors = colors or REG_COLORS[:n_agents]
    spec = dict(W=grid_size, H=grid_size, agents=[dict(color=c) for c in colors],
                view_size=view_size, tile_size=tile_size, view_offset=view_offset,
                see_through_walls=False, max_steps=100, reward_decay=True, ghost_mode=True,
                respawn=False)
    spec.update(kw)
    return spec


def empty_spec(n_agents, grid_size, view_size=7, **kw):
    """EmptyMultiGrid — envs/empty.py:9-16"""
    s = _base(n_agents, grid_size, view_size, **kw)
    W = H = grid_size
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("put", 2, W - 2, H - 2)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    return s


def cluttered_spec(n_agents, grid_size, view_size=7, clutter_density=None, n_clutter=None,
                   randomize_goal=False, **kw):
    """ClutteredMultiGrid — envs/cluttered.py:9-36, including the constructor-time reset that runs
    before n_clutter / randomize_goal exist (random goal, zero clutter)."""
    s = _base(n_agents, grid_size, view_size, **kw)
    W = H = grid_size
    if clutter_density is not None:
        n_clutter = int(clutter_density * (W - 2) * (H - 2))
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    s["gen_ctor"] = [("wall_rect", 0, 0, W, H), ("place", 2, 1, 100)]
    goal = ("place", 2, 1, 100) if randomize_goal else ("put", 2, W - 2, H - 2)
    s["gen_reset"] = [("wall_rect", 0, 0, W, H), goal, ("place", 1, n_clutter, 100)]
    return s


def goalcycle_spec(n_agents, grid_size, view_size=7, clutter_density=None, n_clutter=None,
                   n_bonus_tiles=3, reward=1, penalty=0.0, initial_reward=True,
                   reset_on_mistake=False, **kw):
    """ClutteredGoalCycleEnv — envs/goalcycle.py:9-51 (reward_decay defaults False there; at
    constructor time neither n_bonus_tiles nor n_clutter exist: walls only)."""
    kw.setdefault("reward_decay", False)
    s = _base(n_agents, grid_size, view_size, **kw)
    W = H = grid_size
    if clutter_density is not None:
        n_clutter = int(clutter_density * (W - 2) * (H - 2))
    s["objects"] = [None, WALL] + [
        dict(type="BonusTile", color="yellow", state=b, reward=reward, penalty=penalty, bonus_id=b,
             n_bonus=n_bonus_tiles, initial_reward=initial_reward, reset_on_mistake=reset_on_mistake)
        for b in range(n_bonus_tiles)]
    s["wall_obj"] = 1
    s["gen_ctor"] = [("wall_rect", 0, 0, W, H)]
    s["gen_reset"] = ([("wall_rect", 0, 0, W, H)] + [("place", 2 + b, 1, 100) for b in range(n_bonus_tiles)]
                      + [("place", 1, n_clutter, 100)])
    return s


# BASELINE.json configs + the other registered ids (envs/__init__.py:70-121)
def registered(name):
    table = {
        "MarlGrid-1AgentCluttered15x15-v0": lambda: cluttered_spec(1, 11, 5, n_clutter=30),
        "MarlGrid-3AgentCluttered11x11-v0": lambda: cluttered_spec(3, 11, 7, clutter_density=0.15),
        "MarlGrid-3AgentCluttered15x15-v0": lambda: cluttered_spec(3, 15, 7, clutter_density=0.15),
        "MarlGrid-2AgentEmpty9x9-v0": lambda: empty_spec(2, 9, 7),
        "MarlGrid-3AgentEmpty9x9-v0": lambda: empty_spec(3, 9, 7),
        "MarlGrid-4AgentEmpty9x9-v0": lambda: empty_spec(4, 9, 7),
        "Goalcycle-demo-solo-v0": lambda: goalcycle_spec(1, 13, 7, clutter_density=0.1, n_bonus_tiles=3,
                                                           view_offset=1),
        # BASELINE.json configs[4]: not constructible through register_marl_env (n_agents <= 6)
        "Custom-8AgentCluttered30x30": lambda: cluttered_spec(
            8, 30, 9, clutter_density=0.15,
            colors=["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]),
    }
    return table[name]()
