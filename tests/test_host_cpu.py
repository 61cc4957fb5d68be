"""CPU (-m "not gpu"): host logic of the product package, C-ABI symbol check, scope guards."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "marlgrid_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mg_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    so = os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    L = ctypes.CDLL(so, mode=os.RTLD_NOW)        # RTLD_NOW: also catches undefined internal symbols
    for name in declared:
        assert hasattr(L, name), name
    L.mg_abi_version.restype = ctypes.c_int32
    from marlgrid_amd import _native
    assert L.mg_abi_version() == _native.ABI_VERSION == int(re.search(r"#define MG_ABI_VERSION (\d+)", hdr).group(1))
    assert sorted(_native.SYMBOLS) == declared


def test_struct_sizes_match_the_binding():
    """mg_struct_sizes() = the library's own sizeof of the five ABI structs; _native.lib() refuses to load a
    library whose layouts differ from its ctypes mirrors (what a forgotten field would otherwise turn into
    silently misread launch configs)."""
    from marlgrid_amd import _native as N
    L = N.lib()
    out = (ctypes.c_int32 * 7)()
    assert L.mg_struct_sizes(out) == 7
    assert list(out) == [ctypes.sizeof(t) for t in (N.Config, N.State, N.ObjDesc, N.GenOp, N.GenProgram, N.PlaceTuning, N.PlaceStats)]
    assert L.mg_struct_sizes(None) == -100
    # argument errors of the placement calls are answered on the host, without a device
    assert L.mg_obs_place(None, None, 2, 0, 0.0, 0, None, None, None, None) == -100
    assert L.mg_obs_release(None) == 0 and L.mg_obs_release(ctypes.c_void_p(4096)) == -100      # (not a placed buffer)
    assert L.mg_obs_trim(-1) == 0


def test_integration_stub_executes(monkeypatch):
    """The binding INTEGRATION.md shows a maintainer is code, not prose: extract the marked block, execute it
    against the built library — it asserts the ABI version and every struct size itself — and cross-check its
    structs field by field with the product binding."""
    from marlgrid_amd import _native as N
    N.lib()                                     # (loads libamdhip64 through torch first, as the product does)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- binding-stub[^>]*-->\s*```python\n(.*?)```", md, re.S)
    assert m, "INTEGRATION.md: the binding-stub block is gone"
    monkeypatch.setenv("MARLGRID_HIP_LIB", N.LIB_PATH)
    ns = {}
    exec(compile(m.group(1), "INTEGRATION.md:binding-stub", "exec"), ns)
    for mine, theirs in ((N.Config, ns["MgConfig"]), (N.State, ns["MgState"]), (N.ObjDesc, ns["MgObjDesc"]),
                         (N.GenOp, ns["MgGenOp"]), (N.GenProgram, ns["MgGenProgram"]), (N.PlaceTuning, ns["MgPlaceTuning"]),
                         (N.PlaceStats, ns["MgPlaceStats"])):
        assert ctypes.sizeof(mine) == ctypes.sizeof(theirs)
        a = [(n, getattr(mine, n).offset, getattr(mine, n).size) for n, *_ in mine._fields_]
        b = [(n, getattr(theirs, n).offset, getattr(theirs, n).size) for n, *_ in theirs._fields_]
        assert a == b, theirs.__name__
    # every entry point on the reference's path is bound by the stub
    bound = set(re.findall(r"_L\.(mg_[a-z_0-9]+)\.argtypes", m.group(1))) | {"mg_abi_version", "mg_struct_sizes"}
    for need in ("mg_mt_seed", "mg_reset", "mg_step", "mg_step_render", "mg_render_obs", "mg_encode", "mg_put_obj",
                 "mg_place", "mg_render_frame", "mg_obs_place", "mg_obs_release", "mg_obs_trim"):
        assert need in bound, need


def test_production_library_has_no_measurement_switches():
    """the shipped libmarlgrid_hip.so reads no environment variable and carries none of the A/B variants
    (those live in libmarlgrid_hip_ab.so, built with -DMG_AB_VARIANTS and loaded by tools/ only)"""
    so = os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip.so")
    blob = open(so, "rb").read()
    for needle in (b"MG_RENDER", b"MG_STEP", b"getenv"):
        assert needle not in blob, needle
    L = ctypes.CDLL(so)
    L.mg_build_info.restype = ctypes.c_char_p
    info = L.mg_build_info().decode()
    assert info.startswith("libmarlgrid_hip gfx950 abi6 src-") and "variants" not in info


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "marlgrid_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|from\s+\.+oracle|oracle/|oracle\.|libmgoracle|mgo_",
                                     src), os.path.join(dp, f)
                assert not re.search(r"hostemu|emu_", src), os.path.join(dp, f)      # nor the host test harness


@pytest.mark.parametrize("name", ["MarlGrid-1AgentCluttered15x15-v0", "MarlGrid-3AgentCluttered11x11-v0",
                                  "MarlGrid-3AgentCluttered15x15-v0", "MarlGrid-2AgentEmpty9x9-v0",
                                  "MarlGrid-3AgentEmpty9x9-v0", "MarlGrid-4AgentEmpty9x9-v0",
                                  "Goalcycle-demo-solo-v0"])
def test_scenario_spec_matches_independent_restatement(name):
    from marlgrid_amd.envs import make
    env = make(name, _dry=True)
    a = env.scenario_spec()
    assert a["gen_ctor"] == scenarios.registered(name)["gen_ctor"]
    env.reset()
    a, b = env.scenario_spec(), scenarios.registered(name)
    for k in b:
        assert a[k] == b[k], (k, a[k], b[k])


def test_reject_fn_is_recorded_as_tables():
    """place_obj(reject_fn=) inside `_gen_grid` and agent_spawn_kwargs['reject_fn'] (base.py:690-708): the product
    tabulates the callbacks over their sampling rectangles; its plain-data spec equals the independent restatement
    (rejected cells listed), ops that differ only in their table are not merged, and the launch structs carry them."""
    import product_envs
    env = product_envs.build("Test-2AgentReject9x9", _dry=True)
    env.reset()
    a, b = env.scenario_spec(), scenarios.registered("Test-2AgentReject9x9")
    for k in b:
        assert a[k] == b[k], (k, a[k], b[k])
    template, ops = env._dry_trace
    assert [(o[0], o[1]) for o in ops] == [(1, 5), (3, 1)] and all(o[7] is not None for o in ops)
    t = np.frombuffer(ops[1][7], np.uint8).reshape(9, 9)
    assert sorted(map(tuple, np.argwhere(t))) == [(1, 1), (2, 2), (3, 3), (4, 4)]
    env._host_tables()             # (derives the launch config on the host: what a non-dry env does before a launch)
    assert env._spawn_reject is not None and env._spawn_reject[:, :4].all() and not env._spawn_reject[:, 4:].any()
    # a callback that rejects every cell of the rectangle is a RecursionError at run time, not here; an unknown
    # keyword in agent_spawn_kwargs still is a TypeError, as place_obj(**kw) would raise upstream
    env.agent_spawn_kwargs = dict(reject=lambda p: False)
    with pytest.raises(TypeError):
        env._refresh_cfg(env._host_tables()[0])


def test_fuzz_specs_product_equals_restatement():
    """random constructor knobs (scenarios.fuzz_case): the product's host side derives the same
    plain-data spec (object table, generator program, agent options) as the independent restatement"""
    import product_envs
    for i in range(120):
        name = "Fuzz-%d" % i
        env = product_envs.build(name, _dry=True)
        env.reset()
        a, b = env.scenario_spec(), scenarios.registered(name)
        for k in b:
            assert a[k] == b[k], (name, k, a[k], b[k])


def test_static_edits_after_placements_are_recorded_in_order():
    """`_gen_grid` may edit the layout after a random place_obj (upstream's is free Python): the recorder keeps the
    template for what comes before the first placement and turns every later put_obj / wall helper into fill ops
    (max_tries 0) between the placements; the plain-data spec lists everything in `_gen_grid` order."""
    import product_envs
    name = "Test-2AgentLateStatic10x10"
    env = product_envs.build(name, _dry=True)
    env.reset()
    a, b = env.scenario_spec(), scenarios.registered(name)
    for k in b:
        assert a[k] == b[k], (k, a[k], b[k])
    template, ops = env._dry_trace
    assert template[0].all() and template[:, 0].all() and template[8, 8] == 0       # outer walls only: the goal comes later
    kinds = [(op[0], op[1], op[2]) + tuple(op[3:7]) for op in ops]
    wall, goal = env.obj_reg.find(__import__("marlgrid_amd").objects.Wall()), 2
    assert kinds[0] == (wall, 6, 100, 0, 0, 10, 10)                                 # six random walls first
    assert kinds[1] == (goal, 1, 0, 8, 8, 9, 9)                                     # put_obj(Goal) as a 1 x 1 fill
    assert kinds[2] == (wall, 1, 0, 2, 5, 8, 6)                                     # horz_wall(2, 5, 6)
    assert kinds[3] == (0, 1, 0, 3, 5, 4, 6)                                        # put_obj(None): a gap
    assert kinds[4] == (goal, 1, 100, 1, 1, 4, 4)                                   # a placement after the edits
    assert [k[2] for k in kinds[5:]] == [0, 0, 0, 0] and len(ops) == 9              # wall_rect: four fills


@pytest.mark.parametrize("ts", [5, 8, 11, 32])
def test_atlas_builder_against_reference_tiles(ts):
    """product atlas (marlgrid_amd/rendering.py) == tiles captured from the reference"""
    from marlgrid_amd import objects as PO
    from marlgrid_amd import rendering
    g = np.load(os.path.join(GOLD, "atlas.npz"))
    colors = [str(c) for c in g["colors"]]
    objs = [None, PO.Wall(), PO.Goal(color="green", reward=1), PO.Box("yellow"), PO.Door("yellow", 1),
            PO.Door("yellow", 3), PO.BonusTile(color="yellow", reward=1)]
    atlas, slot, n_slots = rendering.build_atlas(objs, colors, ts)
    n_obj, n_ag = len(objs), len(colors)
    assert atlas.shape == (4, 1 + n_obj + n_slots * n_ag * 4, ts, ts, 3)
    assert (atlas[:, 0] == np.array([35, 25, 30])).all()
    for i, key in enumerate(["empty", "wall", "goal", "box_yellow", "door_yellow_open", "door_yellow_locked", "bonus"]):
        assert np.array_equal(atlas[0, 1 + i], g["%s_ts%d" % (key, ts)]), key
    for k in range(n_ag):
        for d in range(4):
            assert np.array_equal(atlas[0, 1 + n_obj + (0 * n_ag + k) * 4 + d], g["agent_ts%d" % ts][k, d])
            assert np.array_equal(atlas[0, 1 + n_obj + (slot[2] * n_ag + k) * 4 + d], g["goal_blend_ts%d" % ts][k, d])
    # orientations are the reference's rotate_grid of orientation 0
    t = atlas[0, 1 + n_obj + 3]
    assert np.array_equal(atlas[3, 1 + n_obj + 3], np.moveaxis(t[:, ::-1], 0, 1))
    assert np.array_equal(atlas[1, 1 + n_obj + 3], np.moveaxis(t[::-1, :], 0, 1))
    assert np.array_equal(atlas[2, 1 + n_obj + 3], t[::-1, ::-1])


def test_seed_words_match_golden_states():
    """host hashing (seeding.py) + numpy's own init_by_array == the reference's seeded states"""
    from marlgrid_amd import seeding
    g = np.load(os.path.join(GOLD, "rng.npz"))
    for s, key in list(zip(g["seeds"], g["mt_key"]))[:16] + list(zip(g["special_seeds"], g["special_mt_key"])):
        rs = np.random.RandomState()
        rs.seed(seeding.seed_words(int(s)))
        assert np.array_equal(rs.get_state()[1], key)
    assert seeding.seed_words(1337) == [1606814319, 1720193504]


def test_independent_learners_contract():
    """README.md:21-63 — per-agent fan-out of observations / actions, episode context manager."""
    import torch
    from marlgrid_amd.agents import IndependentLearners, LearningAgent
    log = []

    class A(LearningAgent):
        def __init__(self, v, **k):
            super().__init__(**k)
            self.v = v

        def action_step(self, obs):
            return torch.full((obs.shape[0],), self.v)

        def save_step(self, *tr):
            log.append((self.v, [tuple(x.shape) for x in tr]))

        def start_episode(self):
            log.append(("start", self.v))

        def end_episode(self):
            log.append(("end", self.v))

    ag = IndependentLearners(A(0, color="red"), A(2, color="blue"))
    obs = torch.zeros((5, 2, 35, 35, 3), dtype=torch.uint8)
    with ag.episode():
        act = ag.action_step(obs)
        assert act.shape == (5, 2) and act[:, 1].eq(2).all()
        ag.save_step(obs, act, obs, torch.zeros(5, 2), torch.zeros(5, dtype=torch.bool))
    assert log[0] == ("start", 0) and log[-1] == ("end", 2)
    assert log[2] == (0, [(5, 35, 35, 3), (5,), (5, 35, 35, 3), (5,), (5,)])
    assert len(list(ag)) == 2


def test_constructor_errors():
    from marlgrid_amd.envs import ClutteredMultiGrid
    from marlgrid_amd.agents import GridAgentInterface
    with pytest.raises(ValueError):
        ClutteredMultiGrid(agents=[GridAgentInterface()], grid_size=9, _dry=True)          # neither
    with pytest.raises(ValueError):
        ClutteredMultiGrid(agents=[GridAgentInterface()], grid_size=9, n_clutter=1, clutter_density=.1, _dry=True)
    with pytest.raises(ValueError):
        ClutteredMultiGrid(agents=[object()], grid_size=9, n_clutter=1, _dry=True)
    with pytest.raises(ValueError):
        GridAgentInterface(observation_style="nope")


def test_oversized_configuration_is_rejected_on_the_host():
    """a grid whose per-env scratch cannot fit the obs kernel's LDS budget is read in place since round 6 (the launcher's
    pick says so, and a RuntimeWarning), with 'prestige' agents too (variant 12 of it: their recoloured tiles in LDS beside
    it); what still cannot be rendered — sixteen 'prestige' agents at 32-pixel tiles: 196 KiB of recoloured sprites per env —
    fails loudly at table build time (never a silent fallback)"""
    from marlgrid_amd import _native as N
    from marlgrid_amd import base as PB
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import EmptyMultiGrid
    env = EmptyMultiGrid(agents=[GridAgentInterface(view_tile_size=8)], grid_size=200, _dry=True)
    cfg, _raw, _flat, _atlas = env._host_tables()
    assert N.render_kernel_name(cfg) == ("mg::render_kernel<0, 0, 4, 8, 3>", 3)
    assert 0 < N.lib().mg_render_obs_lds_bytes(ctypes.byref(cfg)) <= 160 * 1024
    PB._GENERIC_WARNED.clear()
    g = env._groups[0]
    g.kernel_name = "mg::render_kernel<0, 0, 4, 8, 3>"
    with pytest.warns(RuntimeWarning, match="read in place"):
        PB._warn_generic_kernel(g, 3, False)
    env = EmptyMultiGrid(agents=[GridAgentInterface(view_tile_size=8, color="prestige")], grid_size=200, _dry=True)
    cfg, _raw, _flat, _atlas = env._host_tables()
    assert N.render_kernel_name(cfg) == ("mg::render_kernel<0, 0, 4, 12, 3>", 3)
    assert 0 < N.lib().mg_render_obs_lds_bytes(ctypes.byref(cfg)) <= 160 * 1024
    env = EmptyMultiGrid(agents=[GridAgentInterface(view_tile_size=32, color="prestige") for _ in range(16)], grid_size=20, _dry=True)
    env._dry = False                      # exercise the host-side budget check only
    env.device = None
    with pytest.raises(NotImplementedError, match="LDS"):
        env._sync_tables()


def test_env_from_config_and_make():
    """env_from_config (envs/__init__.py:58-67) builds the class named in the dict; make() knows the
    registered ids and rejects unknown ones."""
    from marlgrid_amd import envs as E
    cfg = {"env_class": "ClutteredGoalCycleEnv", "grid_size": 13, "max_steps": 250, "clutter_density": 0.15,
           "respawn": True, "ghost_mode": True, "reward_decay": False, "n_bonus_tiles": 3, "initial_reward": True,
           "penalty": -1.5,
           "agents": [{"view_size": 7, "view_offset": 1, "view_tile_size": 11, "observation_style": "rich",
                       "see_through_walls": False, "color": "prestige"}], "_dry": True}
    env = E.env_from_config(cfg, randomize_seed=False)
    assert isinstance(env, E.ClutteredGoalCycleEnv) and env.num_agents == 1 and env.agents[0].color == "prestige"
    assert env.view_offset == 1 and env.tile_size == 11 and env.respawn is True
    env2 = E.env_from_config(cfg)          # randomize_seed=True adds random.randint(0, 1337**2) to the seed
    assert env2.seeds[0] >= 0
    assert len(E.registered_envs) == 7
    with pytest.raises(KeyError):
        E.make("MarlGrid-NoSuchEnv-v0")
    e3 = E.make("MarlGrid-4AgentEmpty9x9-v0", _dry=True, batch_size=5)
    assert e3.num_agents == 4 and e3.batch_size == 5 and len(e3.seeds) == 5 and e3.seeds[0] == 1337


@pytest.mark.parametrize("ts", [8, 11])
def test_atlas_of_every_object_kind_matches_the_oracle(ts):
    """product atlas (numpy, marlgrid_amd/rendering.py) == oracle tiles (C) for every object class,
    including the ones whose upstream sprite code cannot run (Key / Ball / closed Door: NameError;
    Lava / Floor: no usable sprite) — two independent implementations of the same drawing."""
    from marlgrid_amd import objects as PO
    from marlgrid_amd import rendering
    from oracle import oracle as O
    spec = scenarios.objects_zoo_spec(ts)
    objs = [None, PO.Wall(), PO.Goal(color="green", reward=1), PO.Door("blue", 1), PO.Door("blue", 2),
            PO.Door("blue", 3), PO.Key("blue"), PO.Ball("purple"), PO.Lava(), PO.Floor("grey"),
            PO.BonusTile(color="yellow", reward=1), PO.Box("olive")]
    mine = [None if o is None else (o.type, o.color, o.state) for o in objs]
    theirs = [None if o is None else (o["type"], o["color"], o.get("state", 0)) for o in spec["objects"]]
    assert mine == theirs
    colors = [a["color"] for a in spec["agents"]]
    atlas, slot, n_slots = rendering.build_atlas(objs, colors, ts)
    orc = O.OracleEnv(spec, construct=False)
    n_obj, n_ag = len(objs), len(colors)
    for i in range(n_obj):
        assert np.array_equal(atlas[0, 1 + i], orc.tile(i)), (i, objs[i])
        if slot[i] != 0xFF:
            for k in range(n_ag):
                for d in range(4):
                    assert np.array_equal(atlas[0, 1 + n_obj + (slot[i] * n_ag + k) * 4 + d], orc.tile(i, k, d)), (i, k, d)
    # overlappable kinds: empty, Goal, open Door, Lava, Floor, BonusTile
    assert n_slots == 6


def test_object_registry_interface():
    """ObjectRegistry (base.py:19-64): ids are per object *kind* (hash-by-value), key 0 is None"""
    from marlgrid_amd.base import ObjectRegistry
    from marlgrid_amd import objects as PO
    r = ObjectRegistry()
    assert len(r) == 1 and r.contains_key(0) and not r.contains_key(1) and r.get_next_key() == 1
    k = r.add_object(PO.Wall())
    assert k == 1 and r.get_key(PO.Wall()) == 1 and r.contains_object(PO.Wall()) and not r.contains_object(PO.Goal(color="green", reward=1))
    assert isinstance(r.obj_of_key(1), PO.Wall) and r.get_key(None) == 0 and r.get_next_key() == 2
    with pytest.raises(ValueError):
        r.get_key(PO.GridAgent())


def test_render_lds_query():
    """mg_render_obs_lds_bytes: the library reports what its obs kernel needs (the host never re-derives
    the layout): small for the bench config, atlas dropped from LDS for 32-px tiles, small again for a huge grid (read in
    place since round 6, with 'prestige' agents too), > 160 KiB only where no variant fits — sixteen 'prestige' agents' recoloured
    32-pixel sprites"""
    from marlgrid_amd import _native as N
    L = N.lib()

    def need(n, vs, ts, cells, n_tiles, prestige=0, hide=0):
        c = N.Config()
        c.n_agents, c.view_size, c.tile_size, c.cells_stride, c.n_tiles = n, vs, ts, cells, n_tiles
        c.prestige_mask, c.any_hide = prestige, hide
        return L.mg_render_obs_lds_bytes(ctypes.byref(c))
    bench = need(3, 7, 8, 240, 28)
    assert 20 * 1024 < bench < 64 * 1024
    assert need(3, 7, 8, 240, 28, prestige=0b111) > bench                 # recoloured-tile space
    assert need(3, 7, 8, 240, 28, prestige=0b111, hide=1) > need(3, 7, 8, 240, 28, prestige=0b111)
    assert need(2, 3, 32, 96, 24) < 4 * 24 * 32 * 32 * 3                  # atlas (288 KiB) stays in HBM / L2
    assert need(1, 7, 8, 200 * 200, 12) < 64 * 1024                       # the grid-in-place variant
    assert need(1, 7, 8, 200 * 200, 12, prestige=1) < 64 * 1024           # ... with a 'prestige' agent
    assert need(16, 7, 32, 400, 100, prestige=0xFFFF) > 160 * 1024        # rejected by MultiGridEnv
    assert need(0, 7, 8, 240, 28) < 0 and L.mg_render_obs_lds_bytes(None) < 0


def test_export_video_frame_function(tmp_path, monkeypatch):
    """utils/video.py:export_video against marlgrid/utils/video.py:8-36 with a stand-in for the optional moviepy:
    the clip is asked for frames at arbitrary times; frame(t) = X[min(int(t * fps), T - 1)], every pixel blown up
    rescale_factor times (np.kron with ones upstream), duration T / fps, written to the expanded path."""
    import sys
    import types
    import numpy as np
    calls = {}

    class VideoClip(object):
        def __init__(self, make_frame, duration=None):
            calls["make_frame"], calls["duration"] = make_frame, duration

        def write_videofile(self, path, fps=None):
            calls["path"], calls["fps"] = path, fps

    editor = types.ModuleType("moviepy.editor")
    editor.VideoClip = VideoClip
    pkg = types.ModuleType("moviepy")
    pkg.editor = editor
    monkeypatch.setitem(sys.modules, "moviepy", pkg)
    monkeypatch.setitem(sys.modules, "moviepy.editor", editor)
    from marlgrid_amd.utils.video import export_video
    rng = np.random.RandomState(0)
    X = [rng.randint(0, 256, size=(6, 5, 3)).astype(np.uint8) for _ in range(7)]
    out = tmp_path / "sub" / "clip.mp4"
    export_video(X, str(out), fps=10, rescale_factor=3)
    assert calls["path"] == str(out) and calls["fps"] == 10 and abs(calls["duration"] - 0.7) < 1e-12
    assert (tmp_path / "sub").is_dir()
    want = np.kron(np.stack(X), np.ones((1, 3, 3, 1)))              # upstream's rescale
    for t in (0.0, 0.05, 0.1, 0.35, 0.69, 0.7, 5.0):
        f = calls["make_frame"](t)
        assert f.shape == (18, 15, 3) and np.array_equal(f, want[min(int(t * 10), 6)])
    export_video(np.stack(X), str(out), fps=20, rescale_factor=1)   # an array, no rescale
    assert np.array_equal(calls["make_frame"](0.26), X[5])


def test_shard_pipeline_arguments():
    """ShardPipeline refuses a batch that does not split evenly — before it touches a device"""
    from marlgrid_amd.sharding import ShardPipeline, shard_seeds
    with pytest.raises(ValueError):
        ShardPipeline(lambda **kw: None, 7, parts=2)
    with pytest.raises(ValueError):
        ShardPipeline(lambda **kw: None, 8, parts=0)
    # the seeds a part gets are those of its slice of the one big env
    assert shard_seeds(1337, 8, 1, 2) == [1341, 1342, 1343, 1344]


def test_late_static_edits_are_rectangle_fills_and_visible_to_get():
    """`_gen_grid` edits after the first place_obj (upstream's `_gen_grid` is free Python): a wall helper drawn with
    another type than Wall is a handful of rectangle fills in the reset program, not an op per cell; single cells put one
    after the other along a row are ONE fill; grid.get() inside `_gen_grid` sees what `_gen_grid` drew, late edits
    included; the spec lists the cells as the oracle's program wants them; and the limit's message says what it counts."""
    from marlgrid_amd.base import MultiGridEnv, MultiGrid
    from marlgrid_amd.objects import Wall, Goal, Floor
    seen = {}

    class Late(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.wall_rect(0, 0, width, height)
            self.place_obj(Goal(color="green", reward=1), max_tries=100)
            self.grid.horz_wall(2, 3, 5, obj_type=Floor)               # after a placement, not a Wall: one fill
            for x in range(2, 7):
                self.grid.set(x, 5, Wall())                            # five single cells in a row: one fill
            self.grid.wall_rect(2, 7, 4, 2, obj_type=Floor)            # 4 x 2: top and bottom row adjacent -> merged
            seen["floor"] = self.grid.get(3, 3)
            seen["wall"] = self.grid.get(6, 5)
            seen["border"] = self.grid.get(0, 0)
            seen["none"] = self.grid.get(4, 4)
            self.place_agents = None

    # (grid.get(4, 4): an empty cell inside the goal's sampling rectangle — the recorder answers "empty" and says that it
    # cannot know: upstream's get() there sees what THIS env's earlier place_obj drew)
    with pytest.warns(RuntimeWarning, match=r"grid.get\(4, 4\) inside _gen_grid reads a cell an earlier random place_obj"):
        env = Late(agents=[dict(color="red")], grid_size=11, _dry=True)
    template, ops = env._dry_trace
    fills = [o for o in ops if o[2] == 0]
    assert [(o[3], o[4], o[5], o[6]) for o in fills][:2] == [(2, 3, 7, 4), (2, 5, 7, 6)]
    assert len(fills) <= 5 and len(ops) <= 7
    assert isinstance(seen["floor"], Floor) and isinstance(seen["wall"], Wall) and isinstance(seen["border"], Wall)
    assert seen["none"] is None
    prog = env.scenario_spec()["gen_ctor"]
    puts = [g for g in prog if g[0] == "put"]
    assert len(puts) == 5 + 5 + 8                                       # horz_wall cells, the row of Walls, the 4 x 2 ring

    class TooMany(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.place_obj(Goal(color="green", reward=1), max_tries=100)
            for k in range(40):
                self.grid.set(1 + (k * 2) % 9, 1 + (k * 4) // 9 * 2 % 9, Wall() if k % 2 else Floor())

    # forty-one ops: more than the 32 a launch struct held until round 5 — the program lives in device memory now
    many = TooMany(agents=[dict(color="red")], grid_size=11, _dry=True)
    assert len(many._dry_trace[1]) > 32
    from marlgrid_amd import _native as N
    old = N.MAX_GEN
    N.MAX_GEN = 8                                        # (the sanity bound's message, without recording a thousand ops)
    try:
        with pytest.raises(NotImplementedError, match=r"1 groups of random placements and \d+ rectangle fills"):
            TooMany(agents=[dict(color="red")], grid_size=11, _dry=True)
    finally:
        N.MAX_GEN = old


def test_bench_parity_after_timed_replay(tmp_path):
    """bench.py's parity_after_timed leg without a GPU: the engine's lane-per-env bodies (tests/native host build) play the
    bench's part — reset, then global step s takes pool[s % 64] — and the replay child (the oracle) must agree with the
    state they end in; a corrupted record, a shifted action pool and a wrong step count must each be caught"""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
    import hostemu
    import bench
    wl, B, T = "MarlGrid-3AgentCluttered15x15-v0", 70, 450
    seeds = [1337 + b for b in range(B)]
    emu = hostemu.HostEmu(wl, B, seeds, auto_reset=True)
    emu.reset()
    rng = np.random.RandomState(0)
    pool = [rng.randint(0, 7, size=(B, 3)) for _ in range(64)]
    for s in range(T):
        rew, done = emu.step(pool[s % 64])
    ids = bench.parity_ids(B)
    assert ids == [0, 1, 35, 36, 63, 64, 68, 69]
    W, H = emu.env.width, emu.env.height
    snap = {"workload": wl, "ids": np.array(ids), "seeds": np.array([seeds[b] for b in ids], dtype=np.int64),
            "pool": np.stack([a[ids] for a in pool]).astype(np.int32), "steps": T,
            "grid": emu.grid[ids][:, :W * H].reshape(len(ids), W, H), "agents": emu.rec[ids].view(np.int64),
            "step_count": emu.step_count[ids], "mt": emu.mt[ids], "mt_pos": emu.mt_pos[ids],
            "rewards": rew[ids], "done": done[ids]}

    def replay(sn):
        path = str(tmp_path / "snap.npz")
        np.savez(path, **sn)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--parity-replay", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    res = replay(snap)
    assert res["ok"] and res["steps"] == T and res["envs"] == 8 and res["episodes_per_env_min"] >= 4, res
    bad = dict(snap, agents=snap["agents"].copy())
    bad["agents"][3, 1] ^= 1                                # one agent of one env one cell off
    assert not replay(bad)["ok"]
    assert not replay(dict(snap, pool=np.roll(snap["pool"], 1, axis=0)))["ok"]
    assert not replay(dict(snap, steps=T - 1))["ok"]
    # and through the parent-side wrapper (temp file, clean child environment)
    assert bench.parity_after_timed(snap)["ok"]


def test_render_kernel_name_and_generic_warning():
    """mg_render_kernel_name: the launcher's own pick for a configuration (no device access) — what bench.py prints as
    roofline.kernel and rocprofv3 prints as the kernel — and the RuntimeWarning for a configuration that falls off the
    dispatch table onto the fully run-time instantiation."""
    import warnings
    import product_envs
    from marlgrid_amd import _native as N
    from marlgrid_amd import base as MB
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredMultiGrid, make

    def name_of(env):
        cfg, _raw, _flat, _atlas = env._host_tables()
        return N.render_kernel_name(cfg)
    assert name_of(make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, _dry=True)) == ("mg::render_kernel<7, 8, 16, 0, 0>", 0)
    assert name_of(make("MarlGrid-3AgentCluttered15x15-v0", batch_size=1024, _dry=True)) == ("mg::render_kernel<7, 8, 4, 0, 0>", 0)

    def cl(vs, ts, B=8192, color="red"):
        return ClutteredMultiGrid(agents=[GridAgentInterface(color=color, view_size=vs, view_tile_size=ts)], grid_size=15,
                                  n_clutter=5, batch_size=B, _dry=True)
    assert name_of(cl(7, 5)) == ("mg::render_kernel<7, 5, 16, 0, 2>", 0)              # the reference's class defaults: gather
    assert name_of(cl(9, 8)) == ("mg::render_kernel<9, 8, 16, 0, 0>", 0)
    assert name_of(cl(11, 8)) == ("mg::render_kernel<0, 8, 8, 0, 0>", 1)               # run-time view, chunk raster
    assert name_of(cl(5, 6)) == ("mg::render_kernel<5, 0, 16, 0, 0>", 2)               # compile-time view, run-time tile
    assert name_of(cl(11, 6)) == ("mg::render_kernel<0, 0, 8, 0, 0>", 3)               # the fully generic one
    assert name_of(cl(7, 11, color="prestige"))[0] == "mg::render_kernel<7, 11, 12, 9, 2>"
    # the warning, once per configuration
    MB._GENERIC_WARNED.clear()
    g = cl(11, 6)._groups[0]
    g.kernel_name = "mg::render_kernel<0, 0, 8, 0, 0>"
    with pytest.warns(RuntimeWarning, match="fully run-time instantiation.*view_size 3 ... 9 with this view_tile_size"):
        MB._warn_generic_kernel(g, 3, False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        MB._warn_generic_kernel(g, 3, False)          # said once
        MB._warn_generic_kernel(g, 1, False)          # a run-time view alone is not the generic instantiation
