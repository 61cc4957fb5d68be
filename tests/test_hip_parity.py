"""GPU parity tests: the HIP path (through the C ABI, via marlgrid_amd) against
  (a) the committed golden vectors captured from the real reference, and
  (b) the CPU oracle on the same seeded inputs,
bit-exact for grid state / observations / encode, |d reward| <= 1e-6 (float32 output of a float64
computation; the tolerance BASELINE.json's north_star states)."""
import os

import numpy as np
import pytest

import canon
import product_envs
import scenarios
from golden import refstate
from marlgrid_amd import seeding
from oracle import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRAJ = sorted(f[5:-4] for f in os.listdir(GOLD) if f.startswith("traj_"))
REW_TOL = 1e-6


def _pov(obs):
    """what reset()/step() returned, indexable as [env][agent] -> (P, P, 3) uint8: the (B, n, P, P, 3) tensor,
    or — with 'rich' agents or agents that have their own views — the per-agent list of (B, P_k, P_k, 3)
    views / dicts with a 'pov' entry (base.py:459-471)"""
    if isinstance(obs, (list, tuple)):
        per_agent = [(x["pov"] if isinstance(x, dict) else x).cpu().numpy() for x in obs]
        if len({a.shape for a in per_agent}) == 1:
            return np.stack(per_agent, axis=1)
        return [[a[b] for a in per_agent] for b in range(per_agent[0].shape[0])]
    return obs.cpu().numpy()


def _cmp_rich(obs, g, ti, what):
    """'rich' fields against the reference's (NaN in the fixture = field not observed): all three
    exactly — position is the same float64 divide"""
    S = g["rich_position"].shape[0]
    for k, x in enumerate(obs):
        want_keys = {"pov"}
        for key, arr in (("reward", g["rich_reward"]), ("orientation", g["rich_orientation"])):
            if not np.isnan(arr[0, ti, k]):
                want_keys.add(key)
                assert np.array_equal(x[key].cpu().numpy().astype(np.float64), arr[:, ti, k]), (what, k, key)
        if not np.isnan(g["rich_position"][0, ti, k]).any():
            want_keys.add("position")
            got = x["position"].cpu().numpy()
            assert got.dtype == np.float64 and np.array_equal(got, g["rich_position"][:, ti, k]), (what, k)
        assert (set(x) if isinstance(x, dict) else {"pov"}) == want_keys, (what, k)


def _cmp_canon(got, g, prefix, si, t, what):
    for k in canon.KEYS[:-1]:
        want = g[prefix + k][si] if t is None else g[prefix + k][si, t]
        assert np.array_equal(np.asarray(got[k]), np.asarray(want)), "%s: %s\nhip=%r\ngolden=%r" % (
            what, k, got[k], want)


@pytest.mark.parametrize("name", TRAJ)
def test_golden_trajectory(name):
    import torch
    g = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    S, T, n = g["actions"].shape
    F = g["obs_full"].shape[0]
    Fh = g["obs_a0"].shape[0] if "obs_a0" in g.files else 0      # agents with their own views: per-agent arrays
    env = product_envs.build(name, batch_size=S, seeds=g["seeds"], encode_in_step=True)
    # the product's own scenario description must equal the independent one the oracle was pinned with
    spec = scenarios.registered(name)
    ps = env.scenario_spec()
    assert ps["gen_ctor"] == spec["gen_ctor"]
    st = product_envs.canonical(env)
    obs = _pov(env.gen_obs())
    for si in range(S):
        _cmp_canon(st[si], g, "ctor_", si, None, "%s ctor seed %d" % (name, si))
        assert [refstate.crc(x) for x in obs[si]] == list(g["obs_crc_ctor"][si])
    rich = "rich_position" in g.files
    obs = env.reset()
    if rich:
        _cmp_rich(obs, g, 0, "%s reset" % name)
    obs = _pov(obs)
    ps = env.scenario_spec()
    assert ps["gen_reset"] == spec["gen_reset"] and ps["objects"] == spec["objects"]
    st = product_envs.canonical(env)
    for si in range(S):
        _cmp_canon(st[si], g, "reset_", si, None, "%s reset seed %d" % (name, si))
        assert [refstate.crc(x) for x in obs[si]] == list(g["obs_crc_reset"][si])
        if si < F:
            assert np.array_equal(obs[si], g["obs_reset_full"][si])
        if si < Fh:
            assert all(np.array_equal(obs[si][k], g["obs_reset_a%d" % k][si]) for k in range(n))
    for t in range(T):
        o, r, dn, _ = env.step(torch.from_numpy(g["actions"][:, t].astype(np.int64)))
        if rich:
            _cmp_rich(o, g, t + 1, "%s step %d" % (name, t))
        o, r, dn = _pov(o), r.cpu().numpy(), dn.cpu().numpy()
        st = product_envs.canonical(env)
        enc_step = env.grid_encoding.cpu().numpy()
        enc = env.grid.encode().cpu().numpy()
        for si in range(S):
            what = "%s seed %d step %d" % (name, si, t)
            _cmp_canon(st[si], g, "step_", si, t, what)
            assert np.abs(r[si].astype(np.float64) - g["rewards"][si, t]).max() <= REW_TOL, what
            assert bool(dn[si]) == bool(g["ep_done"][si, t]), what
            assert np.array_equal(enc[si], g["encode"][si, t]), what
            assert np.array_equal(enc_step[si], g["encode"][si, t]), what       # (encode_in_step: written by the step's own launch)
            assert [refstate.crc(x) for x in o[si]] == list(g["obs_crc"][si, t]), what
            if si < F:
                assert np.array_equal(o[si], g["obs_full"][si, t]), what
            if si < Fh:
                assert all(np.array_equal(o[si][k], g["obs_a%d" % k][si, t]) for k in range(n)), what
        if g["reset_after"][:, t].any():
            env.reset(env_mask=g["reset_after"][:, t])
    for si in range(S):
        assert seeding.same_stream(env.numpy_rng_state(si), (g["mt_final"][si], g["mt_final_pos"][si])), (name, si)


@pytest.mark.parametrize("name", ["Test-1AgentGoalcycle11x11-prestige-ts11", "Test-2AgentCluttered9x9-offset2-ts5",
                                  "Edge-2AgentGoalcycle9x9-prestige-tile5"])
def test_without_the_ready_made_gather_atlas(name, monkeypatch):
    """the gather instantiations take the atlas in the raster's LDS layout from the host when it is there (MgConfig.
    atlas_gather_off; marlgrid_amd/base.py builds it) — a C host that does not provide one gets the layout built per workgroup,
    as before: the reference's goldens (11-pixel tiles with a 'prestige' agent; 5-pixel tiles) / the oracle on that path"""
    monkeypatch.setenv("MG_NO_GATHER_ATLAS", "1")
    if name.startswith("Test-"):
        test_golden_trajectory(name)
    else:
        test_batch_vs_oracle(name, 300, 90)
    monkeypatch.delenv("MG_NO_GATHER_ATLAS")
    env = product_envs.build(name, batch_size=2)
    assert env._cfg.atlas_gather_off > 0 and env.kernel_name.endswith(", 2>")


def test_rng_seeding_golden():
    g = np.load(os.path.join(GOLD, "rng.npz"))
    seeds = list(g["seeds"]) + [int(s) for s in g["special_seeds"]]
    env = product_envs.build("MarlGrid-2AgentEmpty9x9-v0", batch_size=len(seeds), seeds=seeds)
    env.seed()      # re-seed: the constructor's reset consumed draws
    want = np.concatenate([g["mt_key"], g["special_mt_key"]])
    # the device holds the lazy form + the first look-ahead head (words 0..15 already regenerated):
    # wound back to numpy's form it is init_by_array's output, word for word, at position 624
    assert (env.mt_pos.cpu().numpy() == 16).all()
    for b in range(len(seeds)):
        mt, pos = env.numpy_rng_state(b)
        assert pos == 624 and np.array_equal(mt, want[b]), b
    # and the head is what numpy draws first
    head = env.mt_head.cpu().numpy().view(np.uint32)
    for b in (0, len(seeds) - 1):
        rs = np.random.RandomState()
        rs.set_state(("MT19937", want[b], 624, 0, 0.0))
        assert np.array_equal(head[b], rs.randint(0, 2 ** 32, size=16, dtype=np.uint64).astype(np.uint32))


@pytest.mark.parametrize("name,B,T", [("MarlGrid-3AgentCluttered11x11-v0", 4096, 40),
                                       ("MarlGrid-3AgentCluttered15x15-v0", 1024, 120),
                                       ("Custom-8AgentCluttered30x30", 256, 60),
                                       ("Test-4AgentEmpty5x5-crowded", 512, 80),
                                       ("Test-2AgentCluttered9x9-offset2-ts5", 256, 60),
                                       ("Goalcycle-demo-solo-v0", 256, 120),
                                       ("Test-3AgentCluttered9x9-respawn", 256, 150),
                                       ("Test-3AgentEmpty7x7-spawn-delay", 128, 60),
                                       ("Test-4AgentEmpty5x5-hide", 256, 80),
                                       ("Edge-12AgentCluttered9x9-view3", 64, 60),
                                       ("Edge-2AgentCluttered40x40-view9-off3", 64, 40),
                                       ("Edge-3AgentCluttered13x13-view11", 64, 40),
                                       ("Edge-2AgentEmpty8x8-view5-ts4", 64, 40),
                                       ("Edge-16AgentEmpty6x6-view7", 64, 40),
                                       ("Edge-2AgentCluttered9x9-view5-ts16", 4200, 30),
                                       ("Edge-2AgentCluttered9x9-view3-ts32", 64, 30),
                                       ("Edge-3AgentCluttered13x13-view13-ts8", 64, 30),
                                       ("Edge-2AgentEmpty6x6-view3-ts33", 32, 20),
                                       ("Edge-HumanPlayerConfig", 128, 260),
                                       ("Edge-3AgentCluttered11x11-offset6", 64, 40),
                                       ("MarlGrid-3AgentCluttered11x11-v0", 1, 30),
                                       ("MarlGrid-3AgentCluttered11x11-v0", 3, 30),
                                       ("MarlGrid-3AgentCluttered11x11-v0", 65, 30),
                                       ("MarlGrid-4AgentEmpty9x9-v0", 4097, 20),
                                       ("Edge-3AgentCluttered15x15-default-tiles", 4100, 40),
                                       ("Edge-3AgentCluttered15x15-tile6", 4099, 30),
                                       ("Edge-3AgentCluttered15x15-tile6", 37, 30),
                                       ("Edge-5AgentEmpty9x9-tile5-offset3", 4133, 25),
                                       ("Edge-5AgentEmpty9x9-tile5-offset3", 9, 25),
                                       ("Edge-3AgentCluttered11x11-tile7", 4101, 20),
                                       ("Edge-3AgentCluttered11x11-tile9", 67, 20),
                                       ("Edge-3AgentCluttered11x11-tile10", 4098, 20),
                                       ("Edge-3AgentCluttered11x11-tile11", 4097, 20),
                                       ("Edge-3AgentCluttered11x11-tile11", 11, 20),
                                       ("Edge-3AgentCluttered11x11-tile12", 4103, 20),
                                       ("Edge-3AgentCluttered11x11-tile13", 4099, 20),
                                       ("Edge-3AgentCluttered11x11-view3-tile5", 4105, 20),
                                       ("Edge-3AgentCluttered11x11-view5-tile5", 4102, 20),
                                       ("Edge-3AgentCluttered11x11-view5-tile5", 13, 20),
                                       ("Edge-3AgentCluttered11x11-view9-tile5", 4107, 20),
                                       ("Edge-3AgentCluttered11x11-view9-tile5", 66, 20),
                                       ("Edge-3AgentCluttered11x11-view4-tile5", 4104, 20),
                                       ("Edge-3AgentCluttered11x11-view6-tile5", 4106, 20),
                                       ("Edge-3AgentCluttered11x11-view6-tile5", 10, 20),
                                       ("Edge-3AgentCluttered11x11-view8-tile5", 4108, 20),
                                       ("Edge-3AgentCluttered11x11-view4-tile8", 4109, 20),
                                       ("Edge-3AgentCluttered11x11-view6-tile8", 4110, 20),
                                       ("Edge-3AgentCluttered11x11-view6-tile8", 12, 20),
                                       ("Edge-3AgentCluttered11x11-view8-tile8", 4111, 20),
                                       ("Edge-3AgentCluttered11x11-view5-tile6", 4112, 20),
                                       ("Edge-3AgentCluttered11x11-view9-tile7", 4113, 20),
                                       ("Edge-3AgentCluttered11x11-view3-tile13", 4114, 20),
                                       ("Edge-3AgentCluttered11x11-view6-tile4", 4115, 20),
                                       ("Edge-3AgentCluttered11x11-view8-tile11", 4116, 15),
                                       ("Edge-3AgentCluttered11x11-view4-tile3", 14, 20),
                                       ("Edge-3AgentCluttered15x15-view11-tile5", 4117, 15),
                                       ("Edge-3AgentCluttered15x15-view13-tile5", 4118, 15),
                                       ("Edge-3AgentCluttered15x15-view13-tile5", 17, 15),
                                       ("Edge-3AgentCluttered15x15-view15-tile5", 4119, 12),
                                       ("Test-3AgentCluttered9x9-prestige-mixed", 256, 160),
                                       ("Edge-3AgentCluttered9x9-prestige-mixed-tile5", 4101, 70),
                                       ("Edge-3AgentCluttered9x9-prestige-mixed-tile5", 21, 120),
                                       ("Edge-2AgentGoalcycle9x9-prestige-tile5", 300, 90),
                                       # beyond round 5's limits: 24 agents (4 envs per staged batch, the sequential agent loop,
                                       # iter_order in a scratch column), grids read in place (160 x 160 with crowded spawns and
                                       # hide_item_types, 255 x 255: the largest a uint8 coordinate addresses)
                                       ("Limit-24AgentEmpty20x20-view5", 70, 50),
                                       ("Limit-24AgentEmpty20x20-view5", 4099, 12),
                                       ("Limit-4AgentSpawnRect160x160-hide", 70, 60),
                                       ("Limit-3AgentSpawnRect150x150-prestige", 70, 90),       # ... with 'prestige' agents: variant 12 of it
                                       ("Limit-2AgentEmpty255x255-view9-ts5", 37, 40),
                                       ("Limit-2AgentCluttered128x128", 130, 30),
                                       ("Limit-2AgentCluttered25x25-view21-tile5", 70, 50),
                                       ("Limit-3AgentCluttered33x33-view31-tile4", 40, 40),
                                       ("Limit-2AgentEmpty19x19-view17-tile8", 4100, 20)])
def test_batch_vs_oracle(name, B, T):
    """same seeds, same actions: HIP batch == B oracle envs, every step, full observations."""
    import torch
    seeds = 5000 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    assert np.array_equal(env.gen_obs().cpu().numpy(), orc.gen_obs())
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    rng = np.random.RandomState(11)
    n = env.num_agents
    for t in range(T):
        a = rng.randint(0, 7, size=(B, n))
        o, r, dn, _ = env.step(torch.from_numpy(a))
        o2, r2, dn2, _ = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), o2), "obs step %d" % t
        assert np.abs(r.cpu().numpy().astype(np.float64) - r2).max() <= REW_TOL
        assert np.array_equal(dn.cpu().numpy(), dn2)
        if t % 10 == 0 or dn2.any():
            enc = env.grid.encode().cpu().numpy()
            st = product_envs.canonical(env)
            for b in list(np.nonzero(dn2)[0][:8]) + [0, B - 1]:
                canon.assert_same(st[b], canon.oracle_canonical(orc.envs[b]), "env %d step %d" % (b, t))
                assert np.array_equal(enc[b], orc.envs[b].encode())
        if dn2.any():
            env.reset(env_mask=dn2)
            for b in np.nonzero(dn2)[0]:
                orc.envs[b].reset()


# 0..39 + picked cases: prestige with hide_item_types (115, 124, 167, 202), atlas in global memory
# (371, 399), prestige with the atlas in global memory (102, 181, 380) and with hide (349, 572)
@pytest.mark.parametrize("i", list(range(40)) + [102, 115, 124, 167, 181, 202, 349, 371, 380, 399, 572])
def test_fuzz_vs_oracle(i):
    """random constructor knobs (scenarios.fuzz_case; the oracle is checked against the reference on
    the same cases in test_oracle_vs_reference.py::test_live_fuzz): HIP batch == oracle, every step"""
    test_batch_vs_oracle("Fuzz-%d" % i, 40 + (i % 3) * 33, 70)


@pytest.mark.parametrize("i", list(range(28)))
def test_fuzz_beyond_the_old_limits_vs_oracle(i):
    """scenarios.fuzz_wide_case — up to 32 agents, views up to 31 x 31, grids up to 255 x 255 (every seventh case: the grid
    read in place), hide_item_types, spawn delays, all three scenario classes; the oracle is checked against the LIVE reference
    on the same cases (test_oracle_vs_reference.py::test_live_fuzz_beyond_the_old_limits) —: HIP batch == oracle, every step"""
    test_batch_vs_oracle("FuzzW-%d" % i, 9 + (i % 4) * 20, 45)


def test_agent_objects_expose_batched_state_and_geometry():
    """env.agents[k].pos / .dir / .done / .active / .carrying and the view-geometry helpers
    (agents.py:141-288) are batched views of the device state: checked against the oracle's agents"""
    import torch
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 64
    seeds = 900 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    env.reset(); orc.reset()
    rng = np.random.RandomState(3)
    for t in range(25):
        a = rng.randint(0, 7, size=(B, 3))
        env.step(torch.from_numpy(a)); orc.step(a)
    vs, off = env.view_size, env.view_offset
    for k, ag in enumerate(env.agents):
        pos, d = ag.pos.cpu().numpy(), ag.dir.cpu().numpy()
        exts, front = ag.get_view_exts().cpu().numpy(), ag.front_pos.cpu().numpy()
        rel, inv = ag.relative_coords(5, 4).cpu().numpy(), ag.in_view(5, 4).cpu().numpy()
        for b in range(B):
            st = canon.oracle_canonical(orc.envs[b])
            assert tuple(pos[b]) == tuple(st["pos"][k]) and d[b] == st["dir"][k]
            assert bool(ag.done[b]) == bool(st["done"][k]) and bool(ag.active[b]) == bool(st["active"][k])
            if pos[b][0] < 0:
                continue
            x, y = pos[b]
            fx, fy = [(1, 0), (0, 1), (-1, 0), (0, -1)][d[b]]
            assert tuple(front[b]) == (x + fx, y + fy)
            # the view square contains the agent at get_view_pos() after rotation: its own cell is in view
            tx, ty, bx, by = exts[b]
            assert tx <= x < bx and ty <= y < by and bx - tx == vs and by - ty == vs
            own = ag.relative_coords(int(x), int(y))[b].cpu().numpy()
            assert tuple(own) == ag.get_view_pos()
            assert inv[b] == (tx <= 5 < bx and ty <= 4 < by) and (rel[b][0] >= 0) == inv[b]
    env.check_agent_position_integrity("after 25 steps")
    # MultiGrid.slice / rotate_grid / opacity in torch == what the obs kernel cropped and rotated itself
    cells, shown, vis = env._render(debug=True)
    for k, ag in enumerate(env.agents):
        ex = ag.get_view_exts()
        for dd in range(4):
            m = (ag.dir == dd) & ag.active
            if not bool(m.any()):
                continue
            sub = env.grid.slice(ex[:, 0], ex[:, 1], vs, vs, rot_k=dd + 1)
            assert torch.equal(sub[m], cells[:, k][m]), (k, dd)
    b0 = env.grid.grid[0].cpu().numpy()
    wall = env.obj_reg.find(__import__("marlgrid_amd").objects.Wall())
    assert np.array_equal(env.grid.opacity[0].cpu().numpy(), b0 == wall)
    assert torch.equal(env.grid.rotate_left(4), env.grid.grid) and env.grid.rotate_left(1).shape[1:] == (env.height, env.width)
    env.agent_state[3, 1] |= 0xFF                      # x = 255: off the grid
    with pytest.raises(AssertionError, match="integrity"):
        env.check_agent_position_integrity("corrupted")
    with pytest.raises(AttributeError):
        env.agents[0].prestige
    with pytest.raises(NotImplementedError):
        env.agents[0].sees(1, 1)


def test_stepping_past_done_like_the_reference():
    """the reference keeps stepping after `done` (no auto-reset, base.py:649-653): step_count runs
    past max_steps, done agents stay inactive, the decay factor goes negative."""
    import torch
    name, B = "MarlGrid-3AgentEmpty9x9-v0", 128
    seeds = 300 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    env.reset(); orc.reset()
    rng = np.random.RandomState(8)
    for t in range(140):
        a = rng.randint(0, 3, size=(B, 3))
        o, r, dn, _ = env.step(torch.from_numpy(a))
        o2, r2, dn2, _ = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(dn.cpu().numpy(), dn2)
        assert np.abs(r.cpu().numpy().astype(np.float64) - r2).max() <= REW_TOL
    assert dn2.all() and int(env.step_count.min().item()) == 140


def test_auto_reset_matches_manual():
    import torch
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 512
    seeds = 900 + np.arange(B)
    e1 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True)
    e2 = product_envs.build(name, batch_size=B, seeds=seeds)
    e1.reset(); e2.reset()
    rng = np.random.RandomState(3)
    for t in range(130):
        a = torch.from_numpy(rng.randint(0, 3, size=(B, 3)))
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert torch.equal(d1, d2) and torch.equal(r1, r2)
        if d2.any():
            o2 = e2.reset(env_mask=d2)
        assert torch.equal(o1, o2), t


def test_view_and_visibility_vs_oracle():
    name, B = "MarlGrid-3AgentCluttered15x15-v0", 256
    seeds = 77 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    env.reset(); orc.reset()
    import torch
    rng = np.random.RandomState(5)
    for t in range(25):
        a = rng.randint(0, 3, size=(B, 3))
        env.step(torch.from_numpy(a)); orc.step(a, render=False)
    for k in range(3):
        cells, vis = env.gen_obs_grid(k)
        cells, vis = cells.cpu().numpy(), vis.cpu().numpy()
        for b in range(0, B, 7):
            v2, c2 = orc.envs[b].view(k)
            c2 = np.where(c2 >= 1000, 0, c2)          # oracle reports agent cell-objects; HIP reports base ids
            assert np.array_equal(vis[b], v2), (b, k)
            if orc.envs[b].state()["active"][k]:      # inactive: the reference hands out a blank grid
                assert np.array_equal(cells[b], c2), (b, k)


def _setup_scene(env, sc, spec):
    from marlgrid_amd import objects as PO
    mk = {"Box": lambda o: PO.Box(o["color"]), "Door": lambda o: PO.Door(o["color"], o["state"]),
          "Key": lambda o: PO.Key(o["color"])}
    env.reset()
    # register every object kind of the independent spec in the same id order
    for o in spec["objects"][3:]:
        env.obj_reg.get_key(mk[o["type"]](o))
    assert env.scenario_spec()["objects"] == spec["objects"]
    for k, (x, y, d) in enumerate(sc["agents"]):
        env.set_agent(k, x=x, y=y, dir=d)
    for (oid, x, y) in sc["objects"]:
        env.put_obj(None if oid == 0 else mk[spec["objects"][oid]["type"]](spec["objects"][oid]), x, y)
    for k, oid in sc.get("carrying", {}).items():
        env.set_agent(k, carrying=mk[spec["objects"][oid]["type"]](spec["objects"][oid]))


@pytest.mark.parametrize("sname", sorted(scenarios.interact_scenes()))
def test_interact_golden(sname):
    import torch
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import EmptyMultiGrid
    g = np.load(os.path.join(GOLD, "interact.npz"))
    spec = scenarios.interact_spec()
    sc = scenarios.interact_scenes()[sname]
    env = EmptyMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=8) for c in ("red", "blue")],
                         grid_size=7, batch_size=3, seed=1337)
    _setup_scene(env, sc, spec)
    exc = {"ValueError": ValueError, "TypeError": TypeError, "AssertionError": AssertionError,
           "AttributeError": AttributeError}
    for t, act in enumerate(sc["actions"]):
        err = str(g["%s/error" % sname][t])
        what = "%s step %d" % (sname, t)
        a = torch.tensor([act] * 3)
        if err in exc:
            with pytest.raises(exc[err]):
                env.step(a)             # (strict=True polls for recorded errors: the next call into the env raises)
                env.check_errors()
            break
        o, r, dn, _ = env.step(a)
        st = product_envs.canonical(env)
        for b in range(3):
            for k in canon.KEYS[:-1]:
                assert np.array_equal(np.asarray(st[b][k]), g["%s/step_%s" % (sname, k)][t]), (what, k)
        assert np.array_equal(env.grid.encode().cpu().numpy()[1], g["%s/encode" % sname][t]), what
        if err == "":
            assert np.abs(r.cpu().numpy()[2] - g["%s/rewards" % sname][t]).max() <= REW_TOL
            assert [refstate.crc(x) for x in o.cpu().numpy()[0]] == list(g["%s/obs_crc" % sname][t]), what


def test_wrong_action_shape_asserts():
    import torch
    env = product_envs.build("MarlGrid-2AgentEmpty9x9-v0", batch_size=4)
    with pytest.raises(AssertionError):
        env.step(torch.zeros((4, 3), dtype=torch.int64))


def test_shard_invariance_and_idempotence():
    """env b's trajectory does not depend on which batch / shard holds it (no cross-env coupling),
    and re-rendering is idempotent — size-independent properties checked at a large batch."""
    import torch
    name = "MarlGrid-3AgentCluttered15x15-v0"
    B = 32768
    big = product_envs.build(name, batch_size=B, strict=False)
    sub_ids = np.array([0, 1, 63, 64, 4095, 20000, B - 1])
    small = product_envs.build(name, batch_size=len(sub_ids), seeds=1337 + sub_ids)
    orc = O.OracleBatch(scenarios.registered(name), 1337 + sub_ids)      # the chain to the oracle, closed in this run
    big.reset(); small.reset()
    assert np.array_equal(big.obs[sub_ids].cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(0)
    for t in range(30):
        a = torch.randint(0, 7, (B, 3), generator=g)
        o1, r1, d1, _ = big.step(a)
        o2, r2, d2, _ = small.step(a[sub_ids])
        assert torch.equal(o1[sub_ids].cpu(), o2.cpu())
        assert torch.equal(r1[sub_ids].cpu(), r2.cpu()) and torch.equal(d1[sub_ids].cpu(), d2.cpu())
        o3, r3, d3, _ = orc.step(a[sub_ids].numpy())
        assert np.array_equal(o2.cpu().numpy(), o3) and np.array_equal(d2.cpu().numpy(), d3), t
        assert np.abs(r2.cpu().numpy().astype(np.float64) - r3).max() <= REW_TOL
    first = big.gen_obs().clone()
    assert torch.equal(first, big.gen_obs())
    big.check_errors()
    # every image is made of whole tiles: each 8x8 block is one atlas tile in one orientation
    tiles = first[:64].reshape(64, 3, 7, 8, 7, 8, 3).permute(0, 1, 2, 4, 3, 5, 6).reshape(-1, 192).cpu().numpy()
    atlas = big.atlas.reshape(-1, 192)
    known = {bytes(t) for t in atlas}
    assert all(bytes(t) in known for t in np.unique(tiles, axis=0))


def test_full_frame_render_golden():
    """env.render() — whole grid + highlight + agent-view panels — against images captured from the
    reference's MultiGridEnv.render(mode='rgb_array') (tests/golden/frames.npz)."""
    import torch
    g = np.load(os.path.join(GOLD, "frames.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    for name in names:
        seeds = sorted({int(k.split("/")[1]) for k in g.files if k.startswith(name + "/")})
        steps = sorted({int(k.split("/")[2]) for k in g.files if k.startswith(name + "/") and k.endswith("/full")})
        env = product_envs.build(name, batch_size=len(seeds), seeds=seeds)
        env.reset()
        acts = np.stack([g["%s/%d/actions" % (name, s)] for s in seeds])      # (S, T, n)
        for t in range(max(steps) + 1):
            if t in steps:
                panels = ("%s/%d/%d/nopanels" % (name, seeds[0], t)) not in g.files
                full = env.render(env_ids=list(range(len(seeds))), show_agent_views=panels).cpu().numpy()
                bare = env.render(highlight=False, show_agent_views=False, env_ids=list(range(len(seeds)))).cpu().numpy()
                for si, s in enumerate(seeds):
                    assert np.array_equal(bare[si], g["%s/%d/%d/bare" % (name, s, t)]), (name, s, t, "bare")
                    assert np.array_equal(full[si], g["%s/%d/%d/full" % (name, s, t)]), (name, s, t, "full")
                one = env.render(show_agent_views=panels).cpu().numpy()
                assert np.array_equal(one, full[0])
            env.step(torch.from_numpy(acts[:, t].astype(np.int64)))


def test_full_frame_render_of_grids_beyond_lds_golden():
    """env.render() of grids beyond ~180 x 180 (the frame kernel keeps two bytes per cell in LDS and reads the grid where it
    lives): 150 x 150 with 'prestige' agents, 200 x 200 with hide_item_types, 255 x 255 — against the reference's
    MultiGridEnv.render(mode='rgb_array', tile_size=…, show_agent_views=False): shape, CRC32 of the whole image, the crop
    around agent 0 and the top-left corner (tests/golden/frames_big.npz: the images themselves are up to 12 MB)."""
    import torch
    g = np.load(os.path.join(GOLD, "frames_big.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == 3
    for name in names:
        seed, ts = int(g["%s/seed" % name]), int(g["%s/tile_size" % name])
        acts = g["%s/actions" % name]
        steps = sorted({int(k.split("/")[1]) for k in g.files if k.startswith(name + "/") and k.endswith("/full/crc")})
        env = product_envs.build(name, batch_size=2, seeds=[seed, seed + 1000])
        env.reset()
        for t in range(max(steps) + 1):
            if t in steps:
                for tag, kw in (("full", dict()), ("bare", dict(highlight=False))):
                    img = env.render(tile_size=ts, show_agent_views=False, **kw).cpu().numpy()
                    what = (name, t, tag)
                    assert list(img.shape) == list(g["%s/%d/%s/shape" % (name, t, tag)]), what
                    r0, c0 = (int(v) for v in g["%s/%d/%s/at" % (name, t, tag)])
                    assert np.array_equal(img[r0:r0 + 96, c0:c0 + 96], g["%s/%d/%s/near_agent0" % (name, t, tag)]), what
                    assert np.array_equal(img[:64, :64], g["%s/%d/%s/corner" % (name, t, tag)]), what
                    assert refstate.crc(img) == int(g["%s/%d/%s/crc" % (name, t, tag)]), what
            a = np.stack([acts[t], acts[t]]).astype(np.int64)
            env.step(torch.from_numpy(a))


def test_rich_observations():
    """'rich' observation_style: per-agent dicts like the reference's (base.py:461-471)."""
    import torch
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import EmptyMultiGrid
    ag = [GridAgentInterface(color=c, view_size=7, view_tile_size=8, observation_style="rich", observe_position=True,
                             observe_orientation=True, observe_rewards=True) for c in ("red", "blue")]
    env = EmptyMultiGrid(agents=ag, grid_size=9, batch_size=16)
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 2 and set(obs[0]) == {"pov", "reward", "position", "orientation"}
    o, r, d, _ = env.step(torch.randint(0, 3, (16, 2)))
    st = product_envs.canonical(env)
    for k in range(2):
        assert o[k]["pov"].shape == (16, 56, 56, 3) and torch.equal(o[k]["pov"], env.obs[:, k])
        pos = np.stack([s["pos"][k] for s in st]).astype(np.float32) / np.array([9, 9], np.float32)
        assert np.allclose(o[k]["position"].cpu().numpy(), pos)
        assert np.array_equal(o[k]["orientation"].cpu().numpy(), np.array([s["dir"][k] for s in st]))
        assert (o[k]["reward"] == 0).all()


def test_grid_recorder_roundtrip(tmp_path):
    """video.py:55-178 semantics: a frame BEFORE every step, the final frame when reset() ends the episode,
    auto-saved as frames_<reset_count>/frame_<t>.png; export_both writes <id>_frames/"""
    import torch
    from PIL import Image
    from marlgrid_amd.utils.video import GridRecorder
    env = product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=8)
    rec = GridRecorder(env, save_root=str(tmp_path), max_steps=10, env_index=3, auto_save_videos=False)
    assert rec.video_kwargs == {"fps": 20, "rescale_factor": 1} and rec.max_steps == 11
    rec.recording = True
    rec.reset()
    assert rec.ptr == 0 and rec.reset_count == 1
    shots = []
    for t in range(4):
        shots.append(env.render(env_ids=[3])[0].cpu().numpy())     # the state the action is taken in
        rec.step(torch.randint(0, 3, (8, 3)))
    assert rec.ptr == 4 and all(np.array_equal(rec.frames[t], shots[t]) for t in range(4))
    last = env.render(env_ids=[3])[0].cpu().numpy()
    try:
        rec.export_both("ep0")                                     # frames first, then the video ...
    except ImportError:
        pass                                                       # ... which needs moviepy (optional upstream too)
    assert sorted(os.listdir(str(tmp_path / "ep0_frames"))) == ["frame_%d.png" % i for i in range(4)]
    rec.reset()                                                    # appends the final frame, auto-saves
    path = str(tmp_path / "frames_1")
    assert sorted(os.listdir(path)) == ["frame_%d.png" % i for i in range(5)]
    assert np.array_equal(np.asarray(Image.open(os.path.join(path, "frame_4.png"))), last)
    assert rec.ptr == 0 and rec.reset_count == 2 and rec.last_save == 1


def test_readme_loop_with_independent_learners():
    """README.md:39-63 as written (batched): obs from the previous step must survive the next step."""
    import torch
    from marlgrid_amd.agents import IndependentLearners, LearningAgent
    from marlgrid_amd.envs import ClutteredMultiGrid
    seen = []

    class RandomAgent(LearningAgent):
        def action_step(self, obs):
            return torch.randint(0, 3, (obs.shape[0],), device=obs.device)

        def save_step(self, obs, act, next_obs, rew, done):
            seen.append((obs.clone(), next_obs.clone()))

    agents = IndependentLearners(RandomAgent(color="red", view_tile_size=8), RandomAgent(color="blue", view_tile_size=8))
    env = ClutteredMultiGrid(agents, grid_size=9, n_clutter=4, batch_size=32, max_steps=20)
    obs_array = env.reset()
    with agents.episode():
        for _ in range(25):
            prev = obs_array.clone()
            action_array = agents.action_step(obs_array)
            next_obs_array, reward_array, done, _ = env.step(action_array)
            assert torch.equal(obs_array, prev), "the previous observation was overwritten by step()"
            agents.save_step(obs_array, action_array, next_obs_array, reward_array, done)
            obs_array = next_obs_array
            if done.all():
                obs_array = env.reset()
    assert any(not torch.equal(a, b) for a, b in seen)


def test_state_dict_checkpoint_resume():
    """state is plain tensors: saving and restoring it replays the same trajectory (incl. the RNG)."""
    import torch
    env = product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=64, auto_reset=True)
    env.reset()
    g = torch.Generator().manual_seed(1)
    acts = [torch.randint(0, 7, (64, 3), generator=g) for _ in range(30)]
    for a in acts[:10]:
        env.step(a)
    sd = env.state_dict()
    first = [tuple(x.clone() for x in env.step(a)[:3]) for a in acts[10:]]
    env.load_state_dict(sd)
    again = [tuple(x.clone() for x in env.step(a)[:3]) for a in acts[10:]]
    for (o1, r1, d1), (o2, r2, d2) in zip(first, again):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)


def test_full_headline_batch_properties():
    """BASELINE.json's headline batch (262 144 envs of 3AgentCluttered15x15) on ONE GPU: shard
    invariance against a 7-env instance holding the same global env ids, no runtime errors, and
    every 8x8 block of the rendered images is an atlas tile."""
    import torch
    from marlgrid_amd.envs import make
    B = 262144
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, strict=False, obs_buffers=1)
    ids = np.array([0, 1, 65535, 65536, 131071, 200000, B - 1])
    small = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=len(ids), seeds=1337 + ids)
    orc = O.OracleBatch(scenarios.registered("MarlGrid-3AgentCluttered15x15-v0"), 1337 + ids)   # ... and the oracle on the same ids
    env.reset(); small.reset()
    assert np.array_equal(env.obs[ids].cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(0)
    for t in range(12):
        a = torch.randint(0, 7, (B, 3), generator=g)
        o, r, d, _ = env.step(a)
        o2, r2, d2, _ = small.step(a[ids])
        assert torch.equal(o[ids].cpu(), o2.cpu()) and torch.equal(r[ids].cpu(), r2.cpu())
        assert torch.equal(d[ids].cpu(), d2.cpu())
        o3, r3, d3, _ = orc.step(a[ids].numpy())
        assert np.array_equal(o[ids].cpu().numpy(), o3) and np.array_equal(d[ids].cpu().numpy(), d3), t
        assert np.abs(r[ids].cpu().numpy().astype(np.float64) - r3).max() <= REW_TOL
    env.check_errors()
    tiles = env.obs[-64:].reshape(64, 3, 7, 8, 7, 8, 3).permute(0, 1, 2, 4, 3, 5, 6).reshape(-1, 192).cpu().numpy()
    known = {bytes(x) for x in env.atlas.reshape(-1, 192)}
    assert all(bytes(x) in known for x in np.unique(tiles, axis=0))


def test_error_paths_raise_like_the_reference():
    """RecursionError when rejection sampling runs out of tries (base.py:705-706) and AssertionError
    when an agent's front cell is outside the grid (MultiGrid.get asserts, base.py:154-156)."""
    import torch
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.base import MultiGrid, MultiGridEnv
    from marlgrid_amd.envs import ClutteredMultiGrid
    from marlgrid_amd.objects import Wall

    # 3x3 room: one interior cell, two clutter blocks -> the second placement cannot succeed
    env = ClutteredMultiGrid(agents=[GridAgentInterface(view_tile_size=8)], grid_size=3, n_clutter=0, batch_size=4)
    env.n_clutter = 2
    with pytest.raises(RecursionError):
        env.reset()
    # the oracle agrees
    spec = scenarios.cluttered_spec(1, 3, 7, n_clutter=2)
    orc = O.OracleEnv(spec, seed=1337)
    with pytest.raises(RecursionError):
        orc.reset()

    class NoWalls(MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = MultiGrid((width, height))
            self.grid.horz_wall(0, 0, width, obj_type=Wall)       # top row only

    env = NoWalls(agents=[GridAgentInterface(view_tile_size=8)], grid_size=4, batch_size=8, max_steps=1000)
    env.reset()
    with pytest.raises(AssertionError):
        for _ in range(200):
            env.step(torch.randint(0, 3, (8, 1)))    # raised by a later step(): the error flag is polled
        env.check_errors()


def test_live_place_obj_and_try_place_obj_vs_oracle():
    """env.place_obj / env.try_place_obj outside _gen_grid (mg_place): positions, success flags, state
    and RNG consumption equal the oracle's, then the envs keep stepping identically."""
    import torch
    from marlgrid_amd.objects import Goal, Wall
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 96
    seeds = 4000 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    env.reset(); orc.reset()
    for i in range(4):
        pos = env.place_obj(Wall(), max_tries=100).cpu().numpy()
        want = np.array([e.place_obj(1, max_tries=100) for e in orc.envs])
        assert np.array_equal(pos, want)
    pos = env.place_obj(Goal(color="green", reward=1), top=(2, 3), size=(4, 20), max_tries=100).cpu().numpy()
    assert np.array_equal(pos, np.array([e.place_obj(2, region=(2, 3, 6, 11), max_tries=100) for e in orc.envs]))
    # place_obj(reject_fn=) on the live grids: the callback is tabulated (a rejected draw is a spent try)
    rej = tuple((x, y) for x in range(11) for y in range(11) if (x * 3 + y) % 4 != 1)
    pos = env.place_obj(Wall(), reject_fn=lambda p: (p[0] * 3 + p[1]) % 4 != 1, max_tries=400).cpu().numpy()
    assert np.array_equal(pos, np.array([e.place_obj(1, max_tries=400, reject=rej) for e in orc.envs]))
    assert all((x * 3 + y) % 4 == 1 for x, y in pos)
    pos = env.place_obj(env.agents[2], top=(1, 1), size=(6, 6), reject_fn=lambda p: p[0] > p[1]).cpu().numpy()
    rej2 = tuple((x, y) for x in range(1, 7) for y in range(1, 7) if x > y)
    assert np.array_equal(pos, np.array([e.place_obj(-3, region=(1, 1, 7, 7), reject=rej2) for e in orc.envs]))
    # re-seat agent 1 at random (lift + place, highest rank), and try fixed cells for objects / agent 0
    pos = env.place_obj(env.agents[1]).cpu().numpy()
    assert np.array_equal(pos, np.array([e.place_obj(-2) for e in orc.envs]))
    rng = np.random.RandomState(0)
    for i in range(12):
        xy = rng.randint(0, 11, size=(B, 2))
        ok = env.try_place_obj(Wall(), torch.from_numpy(xy)).cpu().numpy()
        assert np.array_equal(ok, np.array([e.try_place_obj(1, *xy[b]) for b, e in enumerate(orc.envs)]))
        ok = env.try_place_obj(env.agents[0], (int(xy[0, 0]), int(xy[0, 1]))).cpu().numpy()
        assert np.array_equal(ok, np.array([e.try_place_obj(-1, xy[0, 0], xy[0, 1]) for e in orc.envs])), i
    st = product_envs.canonical(env)
    for b in range(B):
        canon.assert_same(st[b], canon.oracle_canonical(orc.envs[b]), "env %d" % b)
    assert np.array_equal(env.gen_obs().cpu().numpy(), orc.gen_obs())
    for t in range(20):
        a = rng.randint(0, 7, size=(B, 3))
        o, r, dn, _ = env.step(torch.from_numpy(a))
        o2, r2, dn2, _ = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(dn.cpu().numpy(), dn2)


@pytest.mark.parametrize("i", [3, 4, 8, 10, 21, 24, 27, 31, 33, 36])
def test_operation_fuzz_vs_oracle(i):
    """random interleaving of step / masked reset / live place_obj / try_place_obj (objects and agents)
    on randomly configured envs (scenarios.fuzz_case): state, observations and RNG stay equal to the
    oracle's after every operation"""
    import torch
    from marlgrid_amd.objects import Wall
    name, B = "Fuzz-%d" % i, 24
    spec = scenarios.registered(name)
    W, H, n = spec["W"], spec["H"], len(spec["agents"])
    seeds = 8800 + 100 * i + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds)
    orc = O.OracleBatch(spec, seeds)
    env.reset(); orc.reset()
    rng = np.random.RandomState(77 + i)
    walls = 0

    def same(what):
        st = product_envs.canonical(env)
        for b in range(B):
            canon.assert_same(st[b], canon.oracle_canonical(orc.envs[b]), "%s env %d" % (what, b))
        assert np.array_equal(env.gen_obs().cpu().numpy(), orc.gen_obs()), what
        for b in (0, B - 1):
            assert seeding.same_stream(env.numpy_rng_state(b), orc.envs[b].mt_state()), what

    for t in range(36):
        op = rng.choice(["step", "step", "step", "reset", "place_wall", "try_wall", "try_agent", "reseat"])
        if op == "step":
            a = rng.randint(0, 7, size=(B, n))
            o, r, dn, _ = env.step(torch.from_numpy(a))
            o2, r2, dn2, _ = orc.step(a)
            assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(dn.cpu().numpy(), dn2), t
            assert np.abs(r.cpu().numpy().astype(np.float64) - r2).max() <= REW_TOL
        elif op == "reset":
            m = rng.rand(B) < 0.4
            env.reset(env_mask=m)
            for b in np.nonzero(m)[0]:
                orc.envs[b].reset()
            walls = 0 if m.all() else walls
        elif op == "place_wall" and walls < 3:       # (kept sparse: rejection sampling must not run dry)
            walls += 1
            pos = env.place_obj(Wall(), max_tries=100).cpu().numpy()
            assert np.array_equal(pos, np.array([e.place_obj(1, max_tries=100) for e in orc.envs])), t
        elif op == "try_wall":
            xy = np.stack([rng.randint(0, W, size=B), rng.randint(0, H, size=B)], axis=1)
            ok = env.try_place_obj(Wall(), torch.from_numpy(xy)).cpu().numpy()
            assert np.array_equal(ok, np.array([e.try_place_obj(1, *xy[b]) for b, e in enumerate(orc.envs)])), t
        elif op == "try_agent":
            k = int(rng.randint(n))
            x, y = int(rng.randint(W)), int(rng.randint(H))
            ok = env.try_place_obj(env.agents[k], (x, y)).cpu().numpy()
            assert np.array_equal(ok, np.array([e.try_place_obj(-(k + 1), x, y) for e in orc.envs])), t
        elif op == "reseat":
            k = int(rng.randint(n))
            pos = env.place_obj(env.agents[k]).cpu().numpy()
            assert np.array_equal(pos, np.array([e.place_obj(-(k + 1)) for e in orc.envs])), t
        same("%s @%d" % (op, t))


def test_step_is_stream_capturable():
    """env.step() makes no allocation, host sync or stream switch: it can be captured into a HIP graph
    (torch.cuda.graph) and replayed — every replay equals the oracle stepping the same actions"""
    import torch
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 48
    seeds = 600 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds, obs_buffers=1, strict=False)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    env.reset(); orc.reset()
    rng = np.random.RandomState(2)
    a = rng.randint(0, 7, size=(B, 3))
    static = torch.from_numpy(a).to(env.device)
    env.step(static); orc.step(a)                  # eager once
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                      # recorded, not executed
        obs, rew, done, _ = env.step(static)
    for t in range(12):
        a = rng.randint(0, 7, size=(B, 3))
        static.copy_(torch.from_numpy(a))
        g.replay()
        o2, r2, d2, _ = orc.step(a)
        assert np.array_equal(obs.cpu().numpy(), o2), t
        assert np.abs(rew.cpu().numpy().astype(np.float64) - r2).max() <= REW_TOL
        assert np.array_equal(done.cpu().numpy(), d2)
    env.check_errors()


def test_objects_zoo_vs_oracle():
    """every object class on one board (Lava ends the episode, Floor / open Door / BonusTile are
    walked over, Ball / Key are carried, Doors toggled and unlocked): HIP == oracle step by step."""
    import torch
    from marlgrid_amd import objects as PO
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import EmptyMultiGrid
    spec = scenarios.objects_zoo_spec(8)
    B = 96
    seeds = 700 + np.arange(B)
    env = EmptyMultiGrid(agents=[GridAgentInterface(color=a["color"], view_size=7, view_tile_size=8) for a in spec["agents"]],
                         grid_size=9, batch_size=B, seeds=seeds, max_steps=400)
    spec["max_steps"] = 400
    orc = O.OracleBatch(spec, seeds)
    env.reset(); orc.reset()
    objs = [None, PO.Wall(), PO.Goal(color="green", reward=1), PO.Door("blue", 1), PO.Door("blue", 2),
            PO.Door("blue", 3), PO.Key("blue"), PO.Ball("purple"), PO.Lava(), PO.Floor("grey"),
            PO.BonusTile(color="yellow", reward=1), PO.Box("olive")]
    board = [(5, 2, 2), (3, 4, 2), (4, 6, 2), (6, 2, 5), (7, 5, 2), (8, 3, 3), (9, 4, 4), (10, 5, 5), (4, 1, 6), (8, 6, 6)]
    for o in objs[3:]:
        env.obj_reg.get_key(o)
    # agents to fixed free cells first: put_obj on an occupied cell evicts the agent (the cell's object is
    # replaced, base.py:655-662) — test_interact_golden's put_*_over_agent scenes cover that
    for k, (x, y, d) in enumerate([(1, 1, 0), (7, 1, 2)]):
        env.set_agent(k, x=x, y=y, dir=d)
        for e in orc.envs:
            e.place_agent_at(k, x, y)
            e.set_dir(k, d)
    for (oid, x, y) in board:                 # (object id, x, y)
        env.put_obj(objs[oid], x, y)
        for e in orc.envs:
            e.put_obj(oid, x, y)
    assert env.scenario_spec()["objects"] == spec["objects"]
    assert np.array_equal(env.gen_obs().cpu().numpy(), orc.gen_obs())
    rng = np.random.RandomState(21)
    for t in range(150):
        a = rng.choice(6, size=(B, 2), p=[.15, .15, .4, .1, .1, .1])       # no Box in front => toggle is safe...
        o, r, dn, _ = env.step(torch.from_numpy(a))
        o2, r2, dn2, _ = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), o2), t
        assert np.abs(r.cpu().numpy().astype(np.float64) - r2).max() <= REW_TOL and np.array_equal(dn.cpu().numpy(), dn2)
    st = product_envs.canonical(env)
    for b in range(0, B, 5):
        canon.assert_same(st[b], canon.oracle_canonical(orc.envs[b]), "env %d" % b)


def test_auto_reset_survives_a_table_rebuild():
    """A new object kind registered on a live auto-reset env rebuilds the device tables; the reset program
    the fused auto-reset replays (and the template tensor it points at) must stay valid — compared with an
    env that is reset by hand after every finished episode."""
    import torch
    from marlgrid_amd.objects import Ball, Key
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 256
    seeds = 300 + np.arange(B)
    e1 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True)
    e2 = product_envs.build(name, batch_size=B, seeds=seeds)
    e1.reset(); e2.reset()
    rng = np.random.RandomState(9)
    junk = []
    for t in range(150):
        if t in (3, 40):       # a kind the registry has not seen: tables + atlas are rebuilt and re-uploaded
            obj = Key("blue") if t == 3 else Ball("red")
            cx, cy = 4, 4 + (t == 40)
            # (only where no agent stands on the cell: put_obj over an agent evicts it, as upstream, and its
            # next forward move raises — test_interact_golden covers that)
            free = ~((e1.agent_pos[..., 0] == cx) & (e1.agent_pos[..., 1] == cy)).any(dim=1)
            for e in (e1, e2):
                e.put_obj(obj, cx, cy, env_mask=free)
            junk = [torch.full((e1.cells_stride,), 0xEE, dtype=torch.uint8, device=e1.device) for _ in range(64)]
        a = torch.from_numpy(rng.randint(0, 3, size=(B, 3)))
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert torch.equal(d1, d2) and torch.equal(r1, r2), t
        if d2.any():
            o2 = e2.reset(env_mask=d2)
        assert torch.equal(o1, o2), t
    assert len(junk) == 64


def test_settings_are_read_every_step():
    """max_steps / respawn / ghost_mode changed between steps take effect on the next step, as in the
    reference (which reads the attributes inside step())"""
    import torch
    env = product_envs.build("MarlGrid-2AgentEmpty9x9-v0", batch_size=32, seeds=np.arange(32))
    env.reset()
    a = torch.zeros((32, 2), dtype=torch.int64)
    for t in range(5):
        _, _, d, _ = env.step(a)
    assert not d.any()
    env.max_steps = 6
    _, _, d, _ = env.step(a)
    assert d.all()
    with pytest.raises(IndexError):
        env.render(env_ids=[0, 32])


@pytest.mark.parametrize("name,B,T", [("MarlGrid-3AgentCluttered15x15-v0", 4100, 130),
                                       ("MarlGrid-3AgentCluttered11x11-v0", 37, 120),
                                       ("Test-3AgentCluttered9x9-respawn", 300, 120),
                                       ("Test-3AgentEmpty7x7-spawn-delay", 64, 60),
                                       ("Test-3AgentSpawnRect9x9", 129, 90),
                                       ("Test-2AgentReject9x9", 130, 90),
                                       ("Test-3AgentCluttered9x9-prestige-mixed", 5000, 60),
                                       ("Edge-3AgentCluttered15x15-default-tiles", 4099, 40),
                                       ("Custom-8AgentCluttered30x30", 64, 40),
                                       ("Edge-16AgentEmpty6x6-view7", 40, 30)])
def test_fused_step_equals_two_launches(name, B, T):
    """mg_step_render (the env step fused in front of the raster: one launch) == mg_step then mg_render_obs,
    every step: observations, rewards, done, state, RNG — with auto-reset on, so resets run in both forms"""
    import torch
    seeds = 31000 + np.arange(B)
    e1 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True, fused_step=True)
    e2 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True, fused_step=False)
    assert torch.equal(e1.reset(), e2.reset())
    rng = np.random.RandomState(4)
    n = e1.num_agents
    for t in range(T):
        a = torch.from_numpy(rng.randint(0, 7, size=(B, n)))
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), t
    for k in ("grid_state", "agent_state", "mt_state", "mt_pos", "mt_head", "step_count_t", "error_t"):
        assert torch.equal(getattr(e1, k), getattr(e2, k)), k
    orc = O.OracleBatch(scenarios.registered(name), seeds[:8])     # and both equal the oracle
    e3 = product_envs.build(name, batch_size=8, seeds=seeds[:8], fused_step=True)
    e3.reset(); orc.reset()
    for t in range(min(T, 40)):
        a = rng.randint(0, 7, size=(8, n))
        o, r, d, _ = e3.step(torch.from_numpy(a))
        o2, r2, d2, _ = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(d.cpu().numpy(), d2), t
        if d2.any():
            e3.reset(env_mask=d2)
            for b in np.nonzero(d2)[0]:
                orc.envs[b].reset()


@pytest.mark.parametrize("shared", [False, True])
def test_agents_with_their_own_views_vs_oracle(shared):
    """agents.py:19-35: every agent carries its own view_size / view_tile_size / view_offset / see_through_walls.
    Agents are rendered group by group (one launch per distinct geometry); reset()/step() return the
    per-agent list the reference returns.  Against the oracle on fresh seeds, incl. gen_obs_grid and render().
    shared: agents 0 and 2 have the same geometry (a two-member group next to a one-member group)."""
    import copy
    import torch
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredMultiGrid
    name, B = "Test-3AgentCluttered9x9-hetero-views", 96
    spec = copy.deepcopy(scenarios.registered(name))
    if shared:
        spec["agents"][2]["view"] = dict(spec["agents"][0]["view"])
    seeds = 7000 + np.arange(B)
    agents = [GridAgentInterface(color=a["color"], view_size=a["view"]["view_size"], view_tile_size=a["view"]["tile_size"],
                                 view_offset=a["view"]["view_offset"], see_through_walls=a["view"]["see_through_walls"])
              for a in spec["agents"]]
    env = ClutteredMultiGrid(agents=agents, grid_size=9, n_clutter=6, max_steps=50, batch_size=B, seeds=seeds,
                             auto_reset=True)
    assert [g.members for g in env._groups] == ([[0, 2], [1]] if shared else [[0], [1], [2]])
    orcs = [O.make_env(spec, seed=int(s)) for s in seeds]
    obs = env.reset()
    assert isinstance(obs, list) and [tuple(o.shape) for o in obs] == [(B, 40, 40, 3), (B, 35, 35, 3), (B, 40, 40, 3)]
    want = [o.reset() for o in orcs]
    for k in range(3):
        assert np.array_equal(obs[k].cpu().numpy(), np.stack([w[k] for w in want])), k
    rng = np.random.RandomState(3)
    for t in range(70):
        a = rng.randint(0, 7, size=(B, 3))
        obs, r, d, _ = env.step(torch.from_numpy(a))
        outs = [o.step(a[b]) for b, o in enumerate(orcs)]
        for b, o in enumerate(orcs):
            if outs[b][2]:
                outs[b] = (o.reset(),) + outs[b][1:]          # auto-reset: the returned obs is the new episode's
        for k in range(3):
            assert np.array_equal(obs[k].cpu().numpy(), np.stack([w[0][k] for w in outs])), (t, k)
        assert np.abs(r.cpu().numpy().astype(np.float64) - np.stack([w[1] for w in outs])).max() <= REW_TOL
        assert np.array_equal(d.cpu().numpy(), np.array([w[2] for w in outs]))
    cells, vis = env.gen_obs_grid(1)
    assert tuple(cells.shape) == (B, 7, 7) and bool(vis[env.agent_active[:, 1]].all())   # agent 1 sees through walls
    img = env.render(env_ids=[0, 5])
    assert img.shape[0] == 2 and img.shape[1] == 9 * 32


def _full_size_properties(name, B, n, vs, ts, steps, **kw):
    """size-independent properties at a BASELINE batch: shard invariance against a 7-env twin holding the same
    global env ids — and, so that the run closes its own chain, the ORACLE on those seven seeds, stepped with the same
    actions (it resets on `done`, as the launch does): observations, rewards, done every step, canonical state and RNG
    at the end —, no runtime errors, every tile-sized block of the last envs' images is an atlas tile"""
    import torch
    env = product_envs.build(name, batch_size=B, strict=False, obs_buffers=1, auto_reset=True, **kw)
    ids = np.array([0, 1, B // 4 - 1, B // 4, B // 2 + 5, B - 2, B - 1])
    small = product_envs.build(name, batch_size=len(ids), seeds=1337 + ids, auto_reset=True)
    orc = O.OracleBatch(scenarios.registered(name), 1337 + ids)
    assert env.view_size == vs and env.tile_size == ts and env.num_agents == n
    o, o2 = env.reset(), small.reset()
    assert torch.equal(o[ids].cpu(), o2.cpu())
    assert np.array_equal(o2.cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(1)
    for t in range(steps):
        a = torch.randint(0, 7, (B, n), generator=g)
        o, r, d, _ = env.step(a)
        o2, r2, d2, _ = small.step(a[ids])
        assert torch.equal(o[ids].cpu(), o2.cpu()) and torch.equal(r[ids].cpu(), r2.cpu()), (name, t)
        assert torch.equal(d[ids].cpu(), d2.cpu())
        o3, r3, d3, _ = orc.step(a[ids].numpy(), auto_reset=True)
        assert np.array_equal(o[ids].cpu().numpy(), o3) and np.array_equal(d[ids].cpu().numpy(), d3), (name, t)
        assert np.abs(r[ids].cpu().numpy().astype(np.float64) - r3).max() <= REW_TOL, (name, t)
    env.check_errors()
    st = product_envs.canonical_arrays(env.scenario_spec(), env.grid.grid[ids].cpu().numpy(), env.agent_state[ids].cpu().numpy(),
                                       env.step_count[ids].cpu().numpy())
    for j, b in enumerate(ids):
        canon.assert_same(st[j], canon.oracle_canonical(orc.envs[j]), "%s env %d" % (name, b))
        assert seeding.same_stream(env.numpy_rng_state(int(b)), orc.envs[j].mt_state()), (name, b)
    tb = ts * ts * 3
    tiles = env.obs[-32:].reshape(32, n, vs, ts, vs, ts, 3).permute(0, 1, 2, 4, 3, 5, 6).reshape(-1, tb).cpu().numpy()
    known = {bytes(x) for x in env.atlas.reshape(-1, tb)}
    assert all(bytes(x) in known for x in np.unique(tiles, axis=0))
    return env


def test_full_size_config2_4agent_empty9x9():
    """BASELINE.json configs[2]: MarlGrid-4AgentEmpty9x9-v0 at B = 65 536 (6.2 GB of observations), library-built
    observation buffers, episodes ending and restarting inside the launch (max_steps 100 is not reached in 12
    steps; agents reaching the goal end episodes)"""
    _full_size_properties("MarlGrid-4AgentEmpty9x9-v0", 65536, 4, 7, 8, 12)


def test_full_size_config4_8agent_cluttered30x30():
    """BASELINE.json configs[4] per GPU: 8 agents, 30x30, view 9, B = 131 072 — the <9, 8, 4>-wave instantiation,
    16.3 GB of observations per buffer"""
    env = _full_size_properties("Custom-8AgentCluttered30x30", 131072, 8, 9, 8, 6)
    assert env.obs.numel() == 131072 * 8 * 72 * 72 * 3


def test_full_size_config1_3agent_cluttered11x11():
    """BASELINE.json configs[1]: MarlGrid-3AgentCluttered11x11-v0 at B = 4 096 with auto_reset, past max_steps
    (every env resets in-launch at least once)"""
    env = _full_size_properties("MarlGrid-3AgentCluttered11x11-v0", 4096, 3, 7, 8, 110)
    assert int(env.step_count.max()) < 100
