"""Plain-data scenario specs for the oracle, restating `marlgrid/envs/*.py` + the registered ids
(`marlgrid/envs/__init__.py:70-121`).  Independent of the product package on purpose: the product
emits its own spec (`MultiGridEnv.scenario_spec()`), and a test asserts the two agree.
"""
REG_COLORS = ["red", "blue", "purple", "orange", "olive", "pink"]   # envs/__init__.py:30

WALL = dict(type="Wall", color="worst", state=0)                    # objects.py:47,280
GOAL = dict(type="Goal", color="green", state=0, reward=1)          # cluttered.py:29, empty.py:12


def _base(n_agents, grid_size, view_size, tile_size=8, view_offset=0, colors=None, **kw):
    colors = colors or REG_COLORS[:n_agents]
    spec = dict(W=grid_size, H=grid_size,
                agents=[dict(color=c, prestige_beta=0.95, prestige_scale=2) if c == "prestige" else dict(color=c)
                        for c in colors],
                view_size=view_size, tile_size=tile_size, view_offset=view_offset,
                see_through_walls=False, max_steps=100, reward_decay=True, ghost_mode=True,
                respawn=False)
    spec.update(kw)
    return spec


def empty_spec(n_agents, grid_size, view_size=7, **kw):
    """EmptyMultiGrid — envs/empty.py:9-16"""
    s = _base(n_agents, grid_size, view_size, **kw)
    W, H = s["W"], s["H"]
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
    W, H = s["W"], s["H"]
    if clutter_density is not None:
        n_clutter = int(clutter_density * (W - 2) * (H - 2))
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    s["gen_ctor"] = [("wall_rect", 0, 0, W, H), ("place", 2, 1, 100)]
    goal = ("place", 2, 1, 100) if randomize_goal else ("put", 2, W - 2, H - 2)
    s["gen_reset"] = [("wall_rect", 0, 0, W, H), goal] + ([("place", 1, n_clutter, 100)] if n_clutter else [])
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
                      + ([("place", 1, n_clutter, 100)] if n_clutter else []))
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


# ---------------------------------------------------------------------------------------------
# how to build the same scenario with the *reference* classes (used only where /root/reference
# exists: golden generation + live-parity tests).  (env_class name, env kwargs); agents are built
# from spec["agents"] + view_size/tile_size/view_offset.
# ---------------------------------------------------------------------------------------------

def ref_recipe(name):
    t = {
        "MarlGrid-1AgentCluttered15x15-v0": ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=30)),
        "MarlGrid-3AgentCluttered11x11-v0": ("ClutteredMultiGrid", dict(grid_size=11, clutter_density=0.15)),
        "MarlGrid-3AgentCluttered15x15-v0": ("ClutteredMultiGrid", dict(grid_size=15, clutter_density=0.15)),
        "MarlGrid-2AgentEmpty9x9-v0": ("EmptyMultiGrid", dict(grid_size=9)),
        "MarlGrid-3AgentEmpty9x9-v0": ("EmptyMultiGrid", dict(grid_size=9)),
        "MarlGrid-4AgentEmpty9x9-v0": ("EmptyMultiGrid", dict(grid_size=9)),
        "Goalcycle-demo-solo-v0": ("ClutteredGoalCycleEnv", dict(grid_size=13, clutter_density=0.1, n_bonus_tiles=3)),
        "Custom-8AgentCluttered30x30": ("ClutteredMultiGrid", dict(grid_size=30, clutter_density=0.15)),
        "Test-3AgentCluttered11x11-noghost": ("ClutteredMultiGrid", dict(grid_size=11, clutter_density=0.15, ghost_mode=False)),
        "Test-4AgentEmpty5x5-crowded": ("EmptyMultiGrid", dict(grid_size=5)),
        "Test-4AgentEmpty5x5-crowded-noghost": ("EmptyMultiGrid", dict(grid_size=5, ghost_mode=False)),
        "Test-4AgentEmpty5x5-ghost0": ("EmptyMultiGrid", dict(grid_size=5, ghost_mode=0)),
        "Test-2AgentCluttered9x9-offset2-ts5": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=8, randomize_goal=True)),
        "Test-2AgentEmpty7x7-see-through": ("EmptyMultiGrid", dict(grid_size=7)),
        "Test-3AgentCluttered9x9-respawn": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=6, respawn=True)),
        "Test-4AgentEmpty5x5-respawn-noghost": ("EmptyMultiGrid", dict(grid_size=5, respawn=True, ghost_mode=False)),
        "Test-3AgentEmpty7x7-spawn-delay": ("EmptyMultiGrid", dict(grid_size=7, max_steps=40)),
        "Test-4AgentEmpty5x5-hide": ("EmptyMultiGrid", dict(grid_size=5)),
        "Test-3AgentCluttered9x9-hide": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=8)),
        "Test-2AgentRegion9x9": ("RegionTestEnv", dict(grid_size=9)),
        "Test-3AgentSpawnRect9x9": ("SpawnRectTestEnv", dict(grid_size=9, respawn=True, max_steps=60,
                                                             agent_spawn_kwargs=dict(top=(1, 1), size=(3, 9), max_tries=500))),
        "Test-2AgentReject9x9": ("RejectTestEnv", dict(grid_size=9, respawn=True, max_steps=60,
                                                       agent_spawn_kwargs=dict(reject_fn=eval(REJECT_SPAWN), max_tries=1000))),
        "Test-3AgentEmpty7x7-rich": ("EmptyMultiGrid", dict(grid_size=7, max_steps=40)),
        "Test-3AgentCluttered9x9-hetero-views": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=6, max_steps=50)),
        "Test-2AgentGoalcycle9x9-prestige": ("ClutteredGoalCycleEnv", dict(grid_size=9, n_clutter=4, n_bonus_tiles=3,
                                                                           penalty=-1.5, max_steps=60)),
        "Test-1AgentGoalcycle11x11-prestige-ts11": ("ClutteredGoalCycleEnv", dict(grid_size=11, clutter_density=0.1,
                                                                                  n_bonus_tiles=3, max_steps=80)),
        "Test-3AgentCluttered9x9-prestige-mixed": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=5, respawn=True)),
        "Edge-3AgentCluttered9x9-prestige-mixed-tile5": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=5, respawn=True)),
        "Edge-2AgentGoalcycle9x9-prestige-tile5": ("ClutteredGoalCycleEnv", dict(grid_size=9, n_clutter=4, n_bonus_tiles=3,
                                                                                 penalty=-1.5, max_steps=60)),
        # oracle-only edge shapes (no golden file): view sizes / tile sizes / agent counts / big grids
        "Edge-12AgentCluttered9x9-view3": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=5)),
        "Edge-2AgentCluttered40x40-view9-off3": ("ClutteredMultiGrid", dict(grid_size=40, clutter_density=0.2)),
        "Edge-3AgentCluttered13x13-view11": ("ClutteredMultiGrid", dict(grid_size=13, n_clutter=20)),
        "Edge-2AgentEmpty8x8-view5-ts4": ("EmptyMultiGrid", dict(grid_size=8)),
        "Edge-16AgentEmpty6x6-view7": ("EmptyMultiGrid", dict(grid_size=6)),
        "Edge-2AgentCluttered9x9-view5-ts16": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=6)),
        "Edge-2AgentCluttered9x9-view3-ts32": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=6)),
        "Edge-3AgentCluttered13x13-view13-ts8": ("ClutteredMultiGrid", dict(grid_size=13, n_clutter=20)),
        "Edge-2AgentEmpty6x6-view3-ts33": ("EmptyMultiGrid", dict(grid_size=6)),
        "Edge-3AgentCluttered15x15-default-tiles": ("ClutteredMultiGrid", dict(grid_size=15, n_clutter=10)),
        "Edge-3AgentCluttered15x15-tile6": ("ClutteredMultiGrid", dict(grid_size=15, n_clutter=10)),
        "Edge-5AgentEmpty9x9-tile5-offset3": ("EmptyMultiGrid", dict(grid_size=9)),
        **{"Edge-3AgentCluttered11x11-tile%d" % ts: ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=9)) for ts in (7, 9, 10, 11, 12, 13)},
        **{"Edge-3AgentCluttered11x11-view%d-tile5" % vs: ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=9)) for vs in (3, 4, 5, 6, 8, 9)},
        **{"Edge-3AgentCluttered11x11-view%d-tile8" % vs: ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=9)) for vs in (4, 6, 8)},
        **{"Edge-3AgentCluttered15x15-view%d-tile5" % vs: ("ClutteredMultiGrid", dict(grid_size=15, n_clutter=12)) for vs in (11, 13, 15)},
        **{"Edge-3AgentCluttered11x11-view%d-tile%d" % vt: ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=9))
           for vt in ((5, 6), (9, 7), (3, 13), (6, 4), (8, 11), (4, 3))},
        "Test-3AgentEmpty7x11-nonsquare": ("EmptyMultiGrid", dict(width=7, height=11)),
        "Test-3AgentCluttered12x6-nonsquare": ("ClutteredMultiGrid", dict(width=12, height=6, n_clutter=7)),
        "Test-2AgentLateStatic10x10": ("LateStaticTestEnv", dict(grid_size=10, respawn=True, max_steps=50)),
        "Limit-24AgentEmpty20x20-view5": ("EmptyMultiGrid", dict(grid_size=20, max_steps=60)),
        "Limit-3Agent100Kinds24x24": ("KindsTestEnv", dict(grid_size=24, max_steps=80)),
        "Limit-2AgentCluttered128x128": ("ClutteredMultiGrid", dict(grid_size=128, n_clutter=600, max_steps=40)),
        "Limit-2AgentCluttered25x25-view21-tile5": ("ClutteredMultiGrid", dict(grid_size=25, n_clutter=90, max_steps=50)),
        "Limit-3AgentCluttered33x33-view31-tile4": ("ClutteredMultiGrid", dict(grid_size=33, n_clutter=160, max_steps=50)),
        "Limit-2AgentEmpty19x19-view17-tile8": ("EmptyMultiGrid", dict(grid_size=19, max_steps=50)),
        "Limit-3AgentCluttered200x200-hide": ("ClutteredMultiGrid", dict(grid_size=200, n_clutter=1500, max_steps=40)),
        "Limit-4AgentSpawnRect160x160-hide": ("SpawnRectTestEnv", dict(grid_size=160, respawn=True, max_steps=40,
                                                                       agent_spawn_kwargs=dict(top=(1, 1), size=(3, 3), max_tries=500))),
        "Limit-3AgentSpawnRect150x150-prestige": ("SpawnRectTestEnv", dict(grid_size=150, respawn=True, max_steps=40, goal_at=(2, 2),
                                                                           agent_spawn_kwargs=dict(top=(1, 1), size=(3, 3), max_tries=500))),
        "Limit-2AgentEmpty255x255-view9-ts5": ("EmptyMultiGrid", dict(grid_size=255, max_steps=30)),
        "Limit-2Agent60Groups16x16": ("GroupsTestEnv", dict(grid_size=16, max_steps=60)),
        "Test-3AgentCluttered9x9-view6": ("ClutteredMultiGrid", dict(grid_size=9, n_clutter=7, max_steps=60)),
        "Test-2AgentEmpty8x8-view4-ts5": ("EmptyMultiGrid", dict(grid_size=8, max_steps=50)),
        # the reference's examples/human_player.py configuration (examples/human_player.py:35-55)
        "Edge-3AgentCluttered11x11-offset6": ("ClutteredMultiGrid", dict(grid_size=11, n_clutter=12)),
        "Edge-HumanPlayerConfig": ("ClutteredGoalCycleEnv", dict(grid_size=13, max_steps=250, clutter_density=0.15,
                                                                 respawn=True, ghost_mode=True, reward_decay=False,
                                                                 n_bonus_tiles=3, initial_reward=True, penalty=-1.5)),
    }
    if name.startswith("FuzzW-"):
        return fuzz_wide_case(int(name[6:]))[1]
    if name.startswith("Fuzz-"):
        return fuzz_case(int(name[5:]))[1]
    return t[name]


def _with_delays(spec, delays):
    for a, d in zip(spec["agents"], delays):
        if d:
            a["spawn_delay"] = d
    return spec


def region_spec():
    """test-only scenario with place_obj(top=, size=): see tests/golden/refstate.py:_region_env_class"""
    s = _base(2, 9, 7)
    W = H = 9
    s["objects"] = [None, WALL, GOAL, dict(type="Door", color="yellow", state=3),
                    dict(type="Door", color="yellow", state=1), dict(type="Door", color="yellow", state=2)]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("vert_wall", W // 2, 0, H - 3), ("put", 2, W - 2, H - 2),
            ("place", 3, 1, 100, 0, 0, W // 2, H), ("place", 1, 3, 50, W // 2 + 1, 2, W, H - 1)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    return s


def spawn_rect_spec():
    """test-only scenario whose `_gen_grid` leaves `agent_spawn_kwargs` alone (the shipped scenarios overwrite
    it with {}: empty.py:15, cluttered.py:35, goalcycle.py:50), so that reset / late spawn / respawn place
    agents in the spawn rectangle (base.py:411, 505, 643): see tests/golden/refstate.py:_spawn_rect_env_class"""
    s = _base(3, 9, 7, respawn=True, max_steps=60)
    W = H = 9
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("put", 2, 2, H - 2), ("place", 1, 4, 100)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    s["agent_spawn"] = dict(top=(1, 1), size=(3, 9), max_tries=500)
    s["agents"][2]["spawn_delay"] = 5
    return s


REJECT_CLUTTER = "lambda pos: (pos[0] + pos[1]) % 2 == 0"      # the callbacks of the reject scenario, as text: the
REJECT_DOOR = "lambda pos: pos[0] == pos[1]"                     # reference-side and product-side twins eval() the same
REJECT_SPAWN = "lambda pos: pos[1] < 4"


def reject_spec():
    """test-only scenario with place_obj(reject_fn=) inside `_gen_grid` and agent_spawn_kwargs['reject_fn']
    (base.py:690-708, 411, 505, 643): clutter only on odd-parity cells, a locked Door off the diagonal of a 4x4
    corner, agents (re)spawning in the lower part of the room only.  See tests/golden/refstate.py:_reject_env_class."""
    s = _base(2, 9, 7, respawn=True, max_steps=60)
    W = H = 9
    s["objects"] = [None, WALL, GOAL, dict(type="Door", color="yellow", state=3),
                    dict(type="Door", color="yellow", state=1), dict(type="Door", color="yellow", state=2)]
    s["wall_obj"] = 1
    even = tuple((x, y) for x in range(W) for y in range(H) if (x + y) % 2 == 0)
    diag = tuple((x, x) for x in range(1, 5))
    prog = [("wall_rect", 0, 0, W, H), ("put", 2, W - 2, H - 2), ("place", 1, 5, 200, 0, 0, W, H, even),
            ("place", 3, 1, 100, 1, 1, 5, 5, diag)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    s["agent_spawn"] = dict(max_tries=1000, reject=tuple((x, y) for x in range(W) for y in range(H) if y < 4))
    return s


def late_static_spec():
    """test-only scenario whose `_gen_grid` edits the layout AFTER random placements and places again after the edits:
    see tests/golden/refstate.py:_late_static_env_class"""
    s = _base(2, 10, 7, respawn=True, max_steps=50)
    W = H = 10
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("place", 1, 6, 100), ("put", 2, W - 2, H - 2), ("horz_wall", 2, H // 2, W - 4),
            ("put", 0, 3, H // 2), ("place", 2, 1, 100, 1, 1, 4, 4), ("wall_rect", W - 4, 1, 3, 3)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    return s


# ---- the limits round 6 lifted: more than 16 agents, more than 64 object kinds, more than 32 ops ----
ALL_COLORS = ["red", "orange", "green", "blue", "cyan", "purple", "yellow", "olive", "grey", "worst", "pink", "white",
              "prestige", "shadow"]                                   # objects.py:11-29, in key order


def kinds_list():
    """the hundred objects of the kinds scenario, in the order its `_gen_grid` puts them (plain data: the reference-side and
    the product-side twin build their own objects from it): (class name, colour, kwargs)"""
    out = [("Box", c, {}) for c in ALL_COLORS]
    out += [("Door", c, dict(state=1)) for c in ALL_COLORS]           # open
    out += [("Door", c, dict(state=3)) for c in ALL_COLORS]           # locked
    out += [("Goal", c, dict(reward=1)) for c in ALL_COLORS] + [("Goal", "green", dict(reward=2)), ("Goal", "red", dict(reward=0.5))]
    out += [("BonusTile", c, dict(reward=1, penalty=-0.5, bonus_id=b, n_bonus=3)) for c in ALL_COLORS for b in range(3)]
    assert len(out) == 100
    return out


def kinds_spec():
    """test-only scenario with a hundred objects of 115 kinds on one 24 x 24 board (ids beyond 64; an atlas of ~1 000 tiles that
    is read in place): see tests/golden/refstate.py:_kinds_env_class.  The object list is in the order the product's registry
    fills (a Door brings its other two states along)."""
    s = _base(3, 24, 7, max_steps=80)
    W = H = 24
    objs = [None, WALL]
    index = {}
    prog = [("wall_rect", 0, 0, W, H)]
    for i, (cls, color, kw) in enumerate(kinds_list()):
        if cls == "Door":
            key = (cls, color, kw["state"])
            if key not in index:
                for st in [kw["state"]] + [x for x in (1, 2, 3) if x != kw["state"]]:
                    index[(cls, color, st)] = len(objs)
                    objs.append(dict(type="Door", color=color, state=st))
        elif cls == "Box":
            key = (cls, color)
            index[key] = len(objs)
            objs.append(dict(type="Box", color=color, state=0))
        elif cls == "Goal":
            key = (cls, color, kw["reward"])
            index[key] = len(objs)
            objs.append(dict(type="Goal", color=color, state=0, reward=kw["reward"]))
        else:
            key = (cls, color, kw["bonus_id"])
            index[key] = len(objs)
            objs.append(dict(type="BonusTile", color=color, state=kw["bonus_id"], reward=kw["reward"], penalty=kw["penalty"],
                             bonus_id=kw["bonus_id"], n_bonus=kw["n_bonus"], initial_reward=True, reset_on_mistake=False))
        prog.append(("put", index[key], 1 + 2 * (i % 11), 1 + 2 * (i // 11)))
    prog.append(("place", 1, 6, 100))
    s["objects"] = objs
    s["wall_obj"] = 1
    s["gen_ctor"], s["gen_reset"] = prog, prog
    return s


def groups_spec():
    """test-only scenario whose `_gen_grid` makes SIXTY random placements of alternating kinds — no two neighbours merge, the
    reset program has sixty ops (a launch struct held 32 until round 5): see tests/golden/refstate.py:_groups_env_class"""
    s = _base(2, 16, 7, max_steps=60)
    W = H = 16
    s["objects"] = [None, WALL] + [dict(type="Box", color=c, state=0) for c in ("red", "blue", "green")]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H)]
    for i in range(60):
        prog.append(("place", 1 if i % 2 == 0 else 2 + (i // 2) % 3, 1, 100))
    s["gen_ctor"], s["gen_reset"] = prog, prog
    return s


def big_spawn_rect_spec(size):
    """the spawn-rectangle scenario on a grid that does not fit LDS: four agents crowded into a 3 x 3 corner of a `size` x `size`
    room (stacks, and with hide_item_types the second agent of a cell), respawning there — the obs kernel's grid-in-place variant
    (tests/golden/refstate.py:_spawn_rect_env_class, base.py:411, 505, 643)"""
    s = _base(4, size, 7, respawn=True, max_steps=40)
    W = H = size
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("put", 2, 2, H - 2), ("place", 1, 4, 100)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    s["agent_spawn"] = dict(top=(1, 1), size=(3, 3), max_tries=500)
    return _with_hide(s, [["Agent"], ["Wall"], [], ["Agent", "Goal"]])


def big_prestige_spec(size):
    """'prestige'-coloured agents on a grid that does not fit LDS (the grid-in-place variant with per-env recoloured tiles): three
    agents — two of them 'prestige' — spawn in a 3 x 3 corner of a `size` x `size` room WITH the goal in it, so that rewards (and
    with them the agents' colours, agents.py:92-119, 141-153) change every few steps; respawn=True"""
    s = _base(3, size, 7, respawn=True, max_steps=40, colors=["prestige", "blue", "prestige"])
    W = H = size
    s["objects"] = [None, WALL, GOAL]
    s["wall_obj"] = 1
    prog = [("wall_rect", 0, 0, W, H), ("put", 2, 2, 2), ("place", 1, 4, 100)]
    s["gen_ctor"], s["gen_reset"] = prog, prog
    s["agent_spawn"] = dict(top=(1, 1), size=(3, 3), max_tries=500)
    return s


def _with_views(spec, views):
    """per-agent view geometry (agents.py:19-35); spec-level view_size / tile_size / ... stay the first agent's"""
    for a, v in zip(spec["agents"], views):
        a["view"] = dict(v)
    v0 = views[0]
    spec.update(view_size=v0["view_size"], tile_size=v0["tile_size"], view_offset=v0["view_offset"],
                see_through_walls=v0["see_through_walls"])
    return spec


def _with_rich(spec, rich):
    """observation_style='rich' for the agents whose entry is a dict of observe_* flags (agents.py:24-31)"""
    for a, r in zip(spec["agents"], rich):
        if r is not None:
            a["rich"] = dict(r)
    return spec


_MANY = ["red", "orange", "green", "blue", "cyan", "purple", "yellow", "olive", "grey", "worst", "pink", "white"]


def _with_hide(spec, hides):
    for a, h in zip(spec["agents"], hides):
        if h:
            a["hide_item_types"] = list(h)
    return spec


_registered_base = registered


def registered(name):   # noqa: F811  (extends the table above with test-only scenarios)
    extra = {
        "Test-3AgentCluttered11x11-noghost": lambda: cluttered_spec(3, 11, 7, clutter_density=0.15, ghost_mode=False),
        "Test-4AgentEmpty5x5-crowded": lambda: empty_spec(4, 5, 5),
        "Test-4AgentEmpty5x5-crowded-noghost": lambda: empty_spec(4, 5, 5, ghost_mode=False),
        "Test-4AgentEmpty5x5-ghost0": lambda: empty_spec(4, 5, 5, ghost_mode=0),
        "Test-2AgentCluttered9x9-offset2-ts5": lambda: cluttered_spec(2, 9, 5, n_clutter=8, randomize_goal=True,
                                                                        tile_size=5, view_offset=2),
        "Test-2AgentEmpty7x7-see-through": lambda: empty_spec(2, 7, 3, see_through_walls=True, tile_size=11),
        "Test-3AgentCluttered9x9-respawn": lambda: cluttered_spec(3, 9, 7, n_clutter=6, respawn=True),
        "Test-4AgentEmpty5x5-respawn-noghost": lambda: empty_spec(4, 5, 5, respawn=True, ghost_mode=False),
        "Test-3AgentEmpty7x7-spawn-delay": lambda: _with_delays(empty_spec(3, 7, 5, max_steps=40), [0, 4, 9]),
        "Test-4AgentEmpty5x5-hide": lambda: _with_hide(empty_spec(4, 5, 5), [["Agent"], ["Goal"], ["Wall", "Goal", "Agent"], []]),
        "Test-3AgentCluttered9x9-hide": lambda: _with_hide(cluttered_spec(3, 9, 7, n_clutter=8), [["Wall"], ["Agent", "Goal"], []]),
        "Test-2AgentRegion9x9": lambda: region_spec(),
        "Test-3AgentSpawnRect9x9": lambda: spawn_rect_spec(),
        "Test-2AgentReject9x9": lambda: reject_spec(),
        "Test-2AgentLateStatic10x10": lambda: late_static_spec(),
        "Limit-24AgentEmpty20x20-view5": lambda: empty_spec(24, 20, 5, colors=[ALL_COLORS[k % 12] for k in range(24)], max_steps=60),
        "Limit-3Agent100Kinds24x24": lambda: kinds_spec(),
        "Limit-2AgentCluttered128x128": lambda: cluttered_spec(2, 128, 7, n_clutter=600, max_steps=40),
        "Limit-2AgentCluttered25x25-view21-tile5": lambda: cluttered_spec(2, 25, 21, n_clutter=90, tile_size=5, view_offset=2, max_steps=50),
        "Limit-3AgentCluttered33x33-view31-tile4": lambda: cluttered_spec(3, 33, 31, n_clutter=160, tile_size=4, max_steps=50),
        "Limit-2AgentEmpty19x19-view17-tile8": lambda: empty_spec(2, 19, 17, max_steps=50),
        "Limit-3AgentCluttered200x200-hide": lambda: _with_hide(cluttered_spec(3, 200, 7, n_clutter=1500, max_steps=40),
                                                               [["Wall"], ["Agent", "Goal"], []]),
        "Limit-4AgentSpawnRect160x160-hide": lambda: big_spawn_rect_spec(160),
        "Limit-3AgentSpawnRect150x150-prestige": lambda: big_prestige_spec(150),
        "Limit-2AgentEmpty255x255-view9-ts5": lambda: empty_spec(2, 255, 9, tile_size=5, max_steps=30),
        "Limit-2Agent60Groups16x16": lambda: groups_spec(),
        # every agent its own view (agents.py:19-35): a 5x5 view at 8 px, a 7x7 view at 5 px looking through walls,
        # a 5x5 view at 8 px again (same group as the first) with the agent one row up
        "Test-3AgentCluttered9x9-hetero-views": lambda: _with_views(
            cluttered_spec(3, 9, 5, n_clutter=6, max_steps=50),
            [dict(view_size=5, tile_size=8, view_offset=0, see_through_walls=False),
             dict(view_size=7, tile_size=5, view_offset=0, see_through_walls=True),
             dict(view_size=5, tile_size=8, view_offset=1, see_through_walls=False)]),
        "Test-3AgentEmpty7x7-rich": lambda: _with_rich(_with_delays(empty_spec(3, 7, 5, max_steps=40), [0, 3, 7]),
                                                       [dict(observe_rewards=True, observe_position=True,
                                                             observe_orientation=True), None,
                                                        dict(observe_position=True)]),
        "Test-2AgentGoalcycle9x9-prestige": lambda: goalcycle_spec(2, 9, 7, n_clutter=4, n_bonus_tiles=3, penalty=-1.5,
                                                                   max_steps=60, colors=["prestige", "prestige"]),
        "Test-1AgentGoalcycle11x11-prestige-ts11": lambda: goalcycle_spec(1, 11, 7, clutter_density=0.1, n_bonus_tiles=3,
                                                                          max_steps=80, colors=["prestige"], tile_size=11,
                                                                          view_offset=1),
        "Test-3AgentCluttered9x9-prestige-mixed": lambda: cluttered_spec(3, 9, 7, n_clutter=5, respawn=True,
                                                                         colors=["prestige", "blue", "prestige"]),
        # 'prestige' agents at GridAgentInterface's default tile size (the gather raster with per-env recoloured tiles)
        "Edge-3AgentCluttered9x9-prestige-mixed-tile5": lambda: cluttered_spec(3, 9, 7, n_clutter=5, respawn=True, tile_size=5,
                                                                               colors=["prestige", "blue", "prestige"]),
        "Edge-2AgentGoalcycle9x9-prestige-tile5": lambda: goalcycle_spec(2, 9, 7, n_clutter=4, n_bonus_tiles=3, penalty=-1.5,
                                                                         max_steps=60, colors=["prestige", "prestige"], tile_size=5),
        "Edge-12AgentCluttered9x9-view3": lambda: cluttered_spec(12, 9, 3, n_clutter=5, colors=_MANY[:12]),
        "Edge-2AgentCluttered40x40-view9-off3": lambda: cluttered_spec(2, 40, 9, clutter_density=0.2, view_offset=3),
        "Edge-3AgentCluttered13x13-view11": lambda: cluttered_spec(3, 13, 11, n_clutter=20),
        "Edge-2AgentEmpty8x8-view5-ts4": lambda: empty_spec(2, 8, 5, tile_size=4),
        "Edge-16AgentEmpty6x6-view7": lambda: empty_spec(16, 6, 7, colors=(_MANY + _MANY)[:16]),
        "Edge-2AgentCluttered9x9-view5-ts16": lambda: cluttered_spec(2, 9, 5, n_clutter=6, tile_size=16),
        "Edge-2AgentCluttered9x9-view3-ts32": lambda: cluttered_spec(2, 9, 3, n_clutter=6, tile_size=32, view_offset=1),
        "Edge-3AgentCluttered13x13-view13-ts8": lambda: cluttered_spec(3, 13, 13, n_clutter=20),
        "Edge-2AgentEmpty6x6-view3-ts33": lambda: empty_spec(2, 6, 3, tile_size=33),
        # README.md:36 `ClutteredMultiGrid(agents, grid_size=15, n_clutter=10)` with default agents (view 7, tile 5)
        "Test-3AgentEmpty7x11-nonsquare": lambda: empty_spec(3, 7, 7, H=11),
        "Test-3AgentCluttered12x6-nonsquare": lambda: cluttered_spec(3, 12, 5, n_clutter=7, H=6),
        "Edge-3AgentCluttered15x15-default-tiles": lambda: cluttered_spec(3, 15, 7, n_clutter=10, tile_size=5,
                                                                          colors=["red", "red", "red"]),
        "Edge-3AgentCluttered11x11-offset6": lambda: cluttered_spec(3, 11, 7, n_clutter=12, view_offset=6),
        # the gather raster's other instantiation (view 7, 6-pixel tiles), and 5-pixel tiles with five viewers: an env's
        # stream is then 18 375 bytes — every env of a wave's run starts at another byte phase
        "Edge-3AgentCluttered15x15-tile6": lambda: cluttered_spec(3, 15, 7, n_clutter=10, tile_size=6),
        "Edge-5AgentEmpty9x9-tile5-offset3": lambda: empty_spec(5, 9, 7, tile_size=5, view_offset=3, colors=_MANY[:5]),
        # ... and its instantiations for 7- .. 12-pixel tiles (13: the first tile size without one — assemble-and-stream)
        **{"Edge-3AgentCluttered11x11-tile%d" % ts: (lambda ts=ts: cluttered_spec(3, 11, 7, n_clutter=9, tile_size=ts))
           for ts in (7, 9, 10, 11, 12, 13)},
        # ... and for the other shipped view sizes at the default 5-pixel tiles
        **{"Edge-3AgentCluttered11x11-view%d-tile5" % vs: (lambda vs=vs: cluttered_spec(3, 11, vs, n_clutter=9, tile_size=5, view_offset=vs // 3))
           for vs in (3, 4, 5, 6, 8, 9)},
        # compile-time views with run-time tile sizes (assemble-and-stream, 16-wave workgroups)
        **{"Edge-3AgentCluttered11x11-view%d-tile%d" % vt: (lambda vt=vt: cluttered_spec(3, 11, vt[0], n_clutter=9, tile_size=vt[1], view_offset=vt[0] // 4))
           for vt in ((5, 6), (9, 7), (3, 13), (6, 4), (8, 11), (4, 3))},
        # the large odd views at the default 5-pixel tiles (gather raster, 8-wave workgroups)
        **{"Edge-3AgentCluttered15x15-view%d-tile5" % vs: (lambda vs=vs: cluttered_spec(3, 15, vs, n_clutter=12, tile_size=5, view_offset=vs // 5))
           for vs in (11, 13, 15)},
        # even views with a compile-time size at the registered 8-pixel tiles (the chunk raster)
        **{"Edge-3AgentCluttered11x11-view%d-tile8" % vs: (lambda vs=vs: cluttered_spec(3, 11, vs, n_clutter=9, view_offset=vs // 3))
           for vs in (4, 6, 8)},
        # EVEN view sizes (agents.py:233-266 is written with view_size // 2: in an even view the agent sits at column
        # view_size // 2 when it faces up or right and one column to the left of it when it faces down or left, while
        # the shadow cast always starts from column view_size // 2 — upstream's geometry, reproduced as it is)
        "Test-3AgentCluttered9x9-view6": lambda: cluttered_spec(3, 9, 6, n_clutter=7, max_steps=60, view_offset=1),
        "Test-2AgentEmpty8x8-view4-ts5": lambda: empty_spec(2, 8, 4, tile_size=5, max_steps=50),
        "Edge-HumanPlayerConfig": lambda: goalcycle_spec(1, 13, 7, clutter_density=0.15, n_bonus_tiles=3, penalty=-1.5,
                                                         initial_reward=True, max_steps=250, respawn=True,
                                                         reward_decay=False, colors=["prestige"], tile_size=11,
                                                         view_offset=1),
    }
    if name in extra:
        return extra[name]()
    if name.startswith("FuzzW-"):
        return fuzz_wide_case(int(name[6:]))[0]
    if name.startswith("Fuzz-"):
        return fuzz_case(int(name[5:]))[0]
    return _registered_base(name)


def fuzz_case(i):
    """(spec, reference recipe) of pseudo-random scenario `i`: every constructor knob of
    MultiGridEnv / GridAgentInterface / the three scenario classes drawn at random (base.py:335-347,
    agents.py:24-60, envs/*.py), kept inside what the reference can construct and draw."""
    import random
    r = random.Random(7700 + i)
    kind = r.choice(["empty", "cluttered", "cluttered", "goalcycle"])
    W = r.randint(5, 14)
    H = W if (kind == "goalcycle" or r.random() < 0.5) else r.randint(5, 14)
    free = (W - 2) * (H - 2)
    n = r.randint(1, max(1, min(8, free // 4)))
    vs = r.choice([3, 5, 7, 7, 9])
    common = dict(ghost_mode=r.random() < 0.6, respawn=r.random() < 0.35, max_steps=r.randint(12, 45))
    if r.random() < 0.5:
        common["reward_decay"] = r.random() < 0.5
    p_prestige = r.choice([0, 0, 0.5])
    colors = [("prestige" if r.random() < p_prestige else r.choice(_MANY)) for _ in range(n)]
    akw = dict(view_size=vs, tile_size=r.choice([3, 4, 5, 6, 7, 8, 8, 9, 11, 12]), view_offset=r.randint(0, vs - 1),
               see_through_walls=r.random() < 0.25, colors=colors)
    size = dict(grid_size=W) if W == H and r.random() < 0.5 else dict(width=W, height=H)
    if kind == "empty":
        spec = empty_spec(n, W, H=H, **akw, **common)
        recipe = ("EmptyMultiGrid", dict(size, **common))
    elif kind == "cluttered":
        extra = dict(n_clutter=r.randint(0, free // 5), randomize_goal=r.random() < 0.4)
        spec = cluttered_spec(n, W, H=H, **extra, **akw, **common)
        recipe = ("ClutteredMultiGrid", dict(size, **extra, **common))
    else:
        extra = dict(n_clutter=r.randint(0, free // 6), n_bonus_tiles=r.randint(1, 4),
                     reward=r.choice([1, 1, 2, 0.5]), penalty=r.choice([0.0, -0.5, -1.5]),
                     initial_reward=r.random() < 0.6, reset_on_mistake=r.random() < 0.4)
        spec = goalcycle_spec(n, W, **extra, **akw, **common)
        recipe = ("ClutteredGoalCycleEnv", dict(grid_size=W, **extra, **common))
    r2 = random.Random(99100 + i)       # (its own stream: every other knob of case i stays what it was)
    if r2.random() < 0.25:
        ev = r2.choice([4, 6, 8])        # (a 2 x 2 view cannot be built upstream: MultiGrid needs >= 3)
        spec["view_size"], spec["view_offset"] = ev, min(spec["view_offset"], ev - 1)
    if r.random() < 0.3:
        _with_delays(spec, [r.choice([0, 0, 1, 3, 7]) for _ in range(n)])
    if r.random() < 0.3:
        types = ["Agent", "Wall", "Goal", "BonusTile"]
        _with_hide(spec, [[t for t in types if r.random() < 0.35] for _ in range(n)])
    return spec, recipe


def fuzz_wide_case(i):
    """(spec, reference recipe) of pseudo-random scenario `i` BEYOND round 5's limits: up to 32 agents,
    views up to 31 x 31, grids up to 60 x 60 and — every seventh case — 130 ... 255 cells a side (the obs kernel's grid-in-place
    variant), small tiles so that the images stay small.  The other knobs as in fuzz_case."""
    import random
    r = random.Random(880000 + i)
    kind = r.choice(["empty", "cluttered", "cluttered", "goalcycle"])
    big = i % 7 == 3
    W = r.randint(130, 255) if big else r.randint(8, 60)
    H = W if (kind == "goalcycle" or r.random() < 0.5) else (r.randint(130, 255) if big else r.randint(8, 60))
    free = (W - 2) * (H - 2)
    n = r.choice([1, 2, 3, 5, 9, 12, 17, 20, 24, 32])
    n = max(1, min(n, free // 6))
    vs = r.choice([3, 7, 10, 13, 16, 17, 19, 22, 25, 28, 31]) if not big else r.choice([5, 7, 9, 17])
    if n > 12:
        vs = min(vs, 9)                       # (many agents AND large views: images of megabytes per env; and LDS)
    common = dict(ghost_mode=r.random() < 0.6, respawn=r.random() < 0.35, max_steps=r.randint(12, 45))
    if r.random() < 0.5:
        common["reward_decay"] = r.random() < 0.5
    colors = [r.choice(_MANY) for _ in range(n)]
    akw = dict(view_size=vs, tile_size=r.choice([3, 4, 5, 5, 6, 8]), view_offset=r.randint(0, vs - 1),
               see_through_walls=r.random() < 0.25, colors=colors)
    size = dict(grid_size=W) if W == H and r.random() < 0.5 else dict(width=W, height=H)
    if kind == "empty":
        spec = empty_spec(n, W, H=H, **akw, **common)
        recipe = ("EmptyMultiGrid", dict(size, **common))
    elif kind == "cluttered":
        extra = dict(n_clutter=r.randint(0, min(free // 5, 2000)), randomize_goal=r.random() < 0.4)
        spec = cluttered_spec(n, W, H=H, **extra, **akw, **common)
        recipe = ("ClutteredMultiGrid", dict(size, **extra, **common))
    else:
        extra = dict(n_clutter=r.randint(0, min(free // 6, 2000)), n_bonus_tiles=r.randint(1, 4),
                     reward=r.choice([1, 1, 2, 0.5]), penalty=r.choice([0.0, -0.5, -1.5]),
                     initial_reward=r.random() < 0.6, reset_on_mistake=r.random() < 0.4)
        spec = goalcycle_spec(n, W, **extra, **akw, **common)
        recipe = ("ClutteredGoalCycleEnv", dict(grid_size=W, **extra, **common))
    if r.random() < 0.3:
        _with_delays(spec, [r.choice([0, 0, 1, 3, 7]) for _ in range(n)])
    if r.random() < 0.3:
        types = ["Agent", "Wall", "Goal", "BonusTile"]
        _with_hide(spec, [[t for t in types if r.random() < 0.35] for _ in range(n)])
    return spec, recipe


ALL_SCENARIOS = [
    "MarlGrid-2AgentEmpty9x9-v0", "MarlGrid-3AgentCluttered11x11-v0", "MarlGrid-4AgentEmpty9x9-v0",
    "MarlGrid-3AgentCluttered15x15-v0", "Custom-8AgentCluttered30x30",
    "MarlGrid-1AgentCluttered15x15-v0", "MarlGrid-3AgentEmpty9x9-v0", "Goalcycle-demo-solo-v0",
    "Test-3AgentCluttered11x11-noghost", "Test-4AgentEmpty5x5-crowded",
    "Test-4AgentEmpty5x5-crowded-noghost", "Test-2AgentCluttered9x9-offset2-ts5",
    "Test-2AgentEmpty7x7-see-through", "Test-3AgentCluttered9x9-respawn", "Test-4AgentEmpty5x5-respawn-noghost",
    "Test-3AgentEmpty7x7-spawn-delay", "Test-4AgentEmpty5x5-hide", "Test-3AgentCluttered9x9-hide",
    "Test-2AgentRegion9x9", "Test-2AgentGoalcycle9x9-prestige", "Test-1AgentGoalcycle11x11-prestige-ts11",
    "Test-3AgentCluttered9x9-prestige-mixed", "Test-4AgentEmpty5x5-ghost0", "Test-3AgentEmpty7x11-nonsquare",
    "Test-3AgentCluttered12x6-nonsquare", "Test-3AgentSpawnRect9x9", "Test-3AgentEmpty7x7-rich",
    "Test-3AgentCluttered9x9-hetero-views", "Test-2AgentReject9x9",
    "Test-3AgentCluttered9x9-view6", "Test-2AgentEmpty8x8-view4-ts5", "Test-2AgentLateStatic10x10",
]


# ---------------------------------------------------------------------------------------------
# hand-built interaction scenes (pickup / drop / toggle / errors; base.py:587-620)
# ---------------------------------------------------------------------------------------------

def interact_spec():
    s = empty_spec(2, 7, 7)
    s["objects"] = [None, WALL, GOAL,
                    dict(type="Box", color="yellow", state=0),            # 3
                    dict(type="Door", color="yellow", state=1),           # 4 open
                    dict(type="Door", color="yellow", state=2),           # 5 closed
                    dict(type="Door", color="yellow", state=3),           # 6 locked
                    dict(type="Key", color="yellow", state=0),            # 7
                    dict(type="Key", color="red", state=0)]               # 8
    return s


def objects_zoo_spec(tile_size=8):
    """every object class on one 9x9 board (sprites the reference cannot draw are defined by the
    drawing its code spells out — see oracle/oracle.py:_sprite)"""
    s = empty_spec(2, 9, 7, tile_size=tile_size, colors=["red", "cyan"])
    s["objects"] = [None, WALL, GOAL,
                    dict(type="Door", color="blue", state=1), dict(type="Door", color="blue", state=2),
                    dict(type="Door", color="blue", state=3), dict(type="Key", color="blue", state=0),
                    dict(type="Ball", color="purple", state=0), dict(type="Lava", color="worst", state=0),
                    dict(type="Floor", color="grey", state=0),
                    dict(type="BonusTile", color="yellow", state=0, reward=1, penalty=-0.1, bonus_id=0, n_bonus=1,
                         initial_reward=True, reset_on_mistake=False),
                    dict(type="Box", color="olive", state=0)]
    return s


def interact_scenes():
    """name -> dict(agents=[(x, y, dir)], objects=[(obj_id, x, y)], carrying={agent: obj_id},
    actions=[[a0, a1], ...]).  Keys / closed doors are kept out of sight where the reference's
    renderer would crash (objects.py:309,370)."""
    L, R, F, PICK, DROP, TOG, DONE = 0, 1, 2, 3, 4, 5, 6
    return {
        "box_pickup_drop": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(3, 3, 3)],
                                actions=[[F, DONE], [PICK, DONE], [PICK, DONE], [F, DONE], [DROP, DONE],
                                         [DROP, DONE], [F, DONE], [PICK, L], [R, DONE], [DROP, DONE],
                                         [L, DONE], [TOG, DONE]]),
        "door_locked_nokey": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(6, 3, 3)],
                                  actions=[[TOG, DONE], [F, DONE], [PICK, DONE], [TOG, DONE]]),
        "door_locked_wrongkey": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(6, 3, 3)], carrying={0: 8},
                                     actions=[[TOG, DONE], [F, DONE]]),
        "door_locked_key": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(6, 3, 3)], carrying={0: 7},
                                actions=[[TOG, DONE], [TOG, DONE], [F, DONE], [TOG, DONE], [F, DONE],
                                         [L, DONE], [L, DONE], [TOG, DONE]]),
        "open_door_two_agents": dict(agents=[(2, 3, 0), (3, 2, 1)], objects=[(4, 3, 3)],
                                     actions=[[F, DONE], [DONE, F], [L, R], [F, DONE], [DONE, F],
                                              [DONE, DONE]]),
        "drop_blocked": dict(agents=[(2, 3, 0), (3, 3, 2)], objects=[], carrying={0: 3},
                             actions=[[DROP, DONE], [L, DONE], [DROP, DONE], [PICK, DONE]]),
        "bad_action": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[], actions=[[9, DONE]]),
        # put_obj onto the cell an agent stands on (base.py:655-662) replaces the cell's object: the agent is in no
        # cell any more (others do not see it, it still turns and looks) — and its next successful forward move
        # raises where upstream does: the assert on a solid object, ValueError from list.remove on an overlappable
        # one, AttributeError on None (:555-559)
        "put_box_over_agent": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(3, 2, 3)],
                                   actions=[[L, DONE], [R, R], [F, DONE]]),
        "put_open_door_over_agent": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(4, 2, 3)],
                                         actions=[[R, DONE], [L, R], [TOG, DONE], [F, DONE]]),
        "put_none_over_agent": dict(agents=[(2, 3, 0), (1, 1, 0)], objects=[(0, 2, 3)],
                                    actions=[[DONE, R], [F, DONE]]),
    }
