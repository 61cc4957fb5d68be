#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/*.npz by EXECUTING THE REAL REFERENCE
(/root/reference, imported through tests/golden/refshim).  Build container only.

    python tests/golden/make_golden.py            # regenerate everything

What is captured (SURVEY.md section 4 fixture plan):
  atlas.npz      — sprite tiles via MultiGrid.render_tile for ts in {5,8,11,32}
  rng.npz        — seeded MT19937 states for seeds 1337..1337+63, raw draws, randint / shuffle
  occlusion.npz  — random transparency grids -> occlude_mask outputs
  traj_<scenario>.npz — seeded trajectories: actions, shuffle order, canonical state, rewards,
                        done, grid.encode(), obs (full for 2 seeds, CRC32 for all), final MT state
  interact.npz   — hand-built pickup / drop / toggle / error scenes
The reference's arrays are data; no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refload  # noqa: E402
import refstate  # noqa: E402
import scenarios  # noqa: E402

TRAJ = {  # scenario -> (n_seeds, n_steps, n_full_obs_seeds)
    "MarlGrid-2AgentEmpty9x9-v0": (8, 120, 2),
    "MarlGrid-3AgentCluttered11x11-v0": (12, 150, 2),
    "MarlGrid-4AgentEmpty9x9-v0": (8, 120, 2),
    "MarlGrid-3AgentCluttered15x15-v0": (16, 200, 2),
    "Custom-8AgentCluttered30x30": (4, 80, 1),
    "MarlGrid-1AgentCluttered15x15-v0": (6, 120, 1),
    "Goalcycle-demo-solo-v0": (6, 150, 1),
    "Test-3AgentCluttered11x11-noghost": (8, 150, 1),
    "Test-4AgentEmpty5x5-crowded": (8, 150, 2),
    "Test-4AgentEmpty5x5-crowded-noghost": (8, 150, 1),
    "Test-2AgentCluttered9x9-offset2-ts5": (6, 120, 2),
    # the gather raster's other geometries (round 5), pinned to the reference's own pixels: 6-pixel tiles (63-lane periods),
    # 11-pixel tiles (four sets of lane constants), five viewers at 5 pixels (every env at another byte phase)
    "Edge-3AgentCluttered15x15-tile6": (6, 100, 2),
    "Edge-3AgentCluttered11x11-tile11": (4, 60, 1),
    "Edge-5AgentEmpty9x9-tile5-offset3": (4, 60, 1),
    "Test-2AgentEmpty7x7-see-through": (4, 60, 2),
    "Test-3AgentCluttered9x9-respawn": (8, 200, 1),
    "Test-4AgentEmpty5x5-respawn-noghost": (8, 200, 1),
    "Test-3AgentEmpty7x7-spawn-delay": (8, 120, 2),
    "Test-4AgentEmpty5x5-hide": (8, 150, 2),
    "Test-3AgentCluttered9x9-hide": (6, 120, 2),
    "Test-2AgentRegion9x9": (6, 100, 1),
    "Test-3AgentSpawnRect9x9": (8, 150, 1),
    "Test-2AgentReject9x9": (8, 150, 1),
    "Test-3AgentEmpty7x7-rich": (6, 90, 2),
    "Test-3AgentCluttered9x9-hetero-views": (6, 100, 2),
    "Test-2AgentGoalcycle9x9-prestige": (8, 120, 2),
    "Test-1AgentGoalcycle11x11-prestige-ts11": (4, 100, 1),
    "Test-3AgentCluttered9x9-prestige-mixed": (8, 150, 2),
    "Test-4AgentEmpty5x5-ghost0": (6, 120, 1),
    "Test-3AgentEmpty7x11-nonsquare": (6, 120, 2),
    "Test-3AgentCluttered12x6-nonsquare": (6, 120, 2),
    "Test-3AgentCluttered9x9-view6": (8, 150, 2),
    "Test-2AgentEmpty8x8-view4-ts5": (6, 100, 2),
    "Test-2AgentLateStatic10x10": (8, 120, 1),
    # the limits lifted in round 6: 24 agents (iter_order beyond 16 nibbles), 115 object kinds (ids beyond 64, the atlas read in
    # place), a reset program of 61 ops (beyond the 32 a launch struct held)
    "Limit-24AgentEmpty20x20-view5": (4, 60, 1),
    "Limit-3Agent100Kinds24x24": (4, 80, 1),
    "Limit-2Agent60Groups16x16": (4, 80, 1),
    "Limit-2AgentCluttered128x128": (3, 60, 1),        # a grid of 16 KiB per env (two of them per wave in LDS: grid + first-agent map)
    # grids that do not fit LDS: the obs kernel reads them in place and searches the agents of a view cell (RM_ == 3)
    "Limit-3AgentCluttered200x200-hide": (2, 50, 1),
    "Limit-4AgentSpawnRect160x160-hide": (3, 60, 1),
    "Limit-3AgentSpawnRect150x150-prestige": (3, 60, 1),   # ... with 'prestige' agents (variant 12 of it), the goal where they spawn
    # views beyond 15 x 15 (a view row is a 32-bit mask: up to 31)
    "Limit-2AgentCluttered25x25-view21-tile5": (4, 70, 1),
    "Limit-3AgentCluttered33x33-view31-tile4": (3, 60, 1),
}
# action distributions: navigation mostly, all 7 ids present — except where the reference cannot go: toggling a Box raises
# TypeError (objects.py:381-382) and a closed Door's sprite NameError (objects.py:370), so the scenarios that hold Boxes and open
# Doors never toggle (pickup / drop of a Box are fine)
ACT_P = {"Limit-3Agent100Kinds24x24": [.18, .18, .44, .08, .07, 0., .05], "Limit-2Agent60Groups16x16": [.18, .18, .44, .08, .07, 0., .05]}
CANON = ("base_enc", "pos", "dir", "active", "done", "carry_enc", "ordinal")


def gen_atlas(m, out):
    from marlgrid.base import MultiGrid
    from marlgrid.objects import Wall, Goal, BonusTile, Box, Door, COLORS
    from marlgrid.agents import GridAgentInterface
    d = {}
    colors = [c for c in COLORS if c not in ("prestige", "shadow")]
    for ts in (5, 8, 11, 32):
        d["empty_ts%d" % ts] = MultiGrid.render_tile(None, tile_size=ts).astype(np.uint8)
        d["wall_ts%d" % ts] = MultiGrid.render_tile(Wall(), tile_size=ts).astype(np.uint8)
        d["goal_ts%d" % ts] = MultiGrid.render_tile(Goal(color="green", reward=1), tile_size=ts).astype(np.uint8)
        d["bonus_ts%d" % ts] = MultiGrid.render_tile(BonusTile(color="yellow", reward=1), tile_size=ts).astype(np.uint8)
        d["box_yellow_ts%d" % ts] = MultiGrid.render_tile(Box("yellow"), tile_size=ts).astype(np.uint8)
        d["door_yellow_open_ts%d" % ts] = MultiGrid.render_tile(Door(color="yellow", state=1), tile_size=ts).astype(np.uint8)
        d["door_yellow_locked_ts%d" % ts] = MultiGrid.render_tile(Door(color="yellow", state=3), tile_size=ts).astype(np.uint8)
        sprites = np.zeros((len(colors), 4, ts, ts, 3), np.uint8)
        blends = np.zeros((len(colors), 4, ts, ts, 3), np.uint8)
        for ci, c in enumerate(colors):
            for dr in range(4):
                a = GridAgentInterface(color=c, view_size=3, view_tile_size=ts)
                a.activate()
                a.dir = dr
                sprites[ci, dr] = MultiGrid.render_tile(a, tile_size=ts)
                g = Goal(color="green", reward=1)
                g.agents.append(a)
                blends[ci, dr] = MultiGrid.render_tile(g, tile_size=ts)
        d["agent_ts%d" % ts] = sprites
        d["goal_blend_ts%d" % ts] = blends
    d["colors"] = np.array(colors)
    np.savez_compressed(out, **d)


def gen_rng(out):
    from gym.utils import seeding
    seeds = np.arange(1337, 1337 + 64)
    keys = np.zeros((64, 624), np.uint32)
    for i, s in enumerate(seeds):
        rng, _ = seeding.np_random(int(s))
        st = rng.get_state()
        assert st[2] == 624
        keys[i] = st[1]
    d = dict(seeds=seeds, mt_key=keys)
    # raw tempered 32-bit draws + bounded draws for a few seeds
    raw = np.zeros((4, 2000), np.uint32)
    for i in range(4):
        rng, _ = seeding.np_random(int(seeds[i]))
        raw[i] = rng.randint(0, 2 ** 32, size=2000, dtype=np.uint32)   # one word per draw
    d["raw_draws"] = raw
    ri = np.zeros((4, 300, 2), np.int64)
    sh = np.zeros((4, 300, 8), np.int64)
    for i in range(4):
        rng, _ = seeding.np_random(int(seeds[i]))
        for t in range(300):
            ri[i, t] = rng.randint((0, 0), (15, 11))
            x = np.arange(8)
            rng.shuffle(x)
            sh[i, t] = x
    d["randint_0_0_15_11"] = ri
    d["shuffle8"] = sh
    # big / special seeds through the hashing path
    special = np.array([0, 1, 2 ** 31, 2 ** 32 + 5, 1337 * 1337, 2 ** 63 + 11], dtype=np.uint64)
    sk = np.zeros((len(special), 624), np.uint32)
    for i, s in enumerate(special):
        rng, _ = seeding.np_random(int(s))
        sk[i] = rng.get_state()[1]
    d["special_seeds"] = special
    d["special_mt_key"] = sk
    np.savez_compressed(out, **d)


def gen_occlusion(m, out):
    from marlgrid.agents import occlude_mask
    rng = np.random.RandomState(7)
    d = {}
    for vs in (3, 5, 7, 9, 11, 2, 4, 6, 8):        # (even sizes appended: the odd ones' draws stay what they were)
        for off in (0, 1, 2):
            if vs - 1 - off < 0:
                continue
            N = 300
            T = rng.rand(N, vs, vs) < rng.choice([0.5, 0.7, 0.85, 0.95], size=(N, 1, 1))
            M = np.zeros((N, vs, vs), bool)
            for i in range(N):
                M[i] = occlude_mask(T[i].copy(), (vs // 2, vs - 1 - off))
            d["T_vs%d_off%d" % (vs, off)] = np.packbits(T)
            d["M_vs%d_off%d" % (vs, off)] = np.packbits(M)
    np.savez_compressed(out, **d)


def gen_traj(name, out):
    spec = scenarios.registered(name)
    recipe = scenarios.ref_recipe(name)
    S, T, F = TRAJ[name]
    n, W, H = len(spec["agents"]), spec["W"], spec["H"]
    P = spec["view_size"] * spec["tile_size"]
    # agents with their own views (agents.py:19-35): observations differ in shape, so the full observations of
    # the first F seeds are kept per agent (obs_a<k>, obs_reset_a<k>) instead of in one stacked array
    hetero = any("view" in a for a in spec["agents"])
    Pk = [a.get("view", spec)["view_size"] * a.get("view", spec)["tile_size"] for a in spec["agents"]]
    Fh = F if hetero else 0
    obs_a = [np.zeros((Fh, T, Pk[k], Pk[k], 3), np.uint8) for k in range(n)]
    obs_reset_a = [np.zeros((Fh, Pk[k], Pk[k], 3), np.uint8) for k in range(n)]
    if hetero:
        F = 0
    seeds = 1337 + np.arange(S)
    arng = np.random.RandomState(4242)
    # mostly navigation, all 7 ids present (pickup/drop/toggle/done are no-ops in these scenes)
    actions = arng.choice(7, size=(S, T, n), p=ACT_P.get(name, [.2, .2, .4, .05, .05, .05, .05])).astype(np.int8)
    d = dict(seeds=seeds, actions=actions)
    rec = {k: [] for k in CANON}
    ctor = {k: [] for k in CANON}
    rst = {k: [] for k in CANON}
    rewards = np.zeros((S, T, n), np.float64)
    ep_done = np.zeros((S, T), bool)
    reset_after = np.zeros((S, T), bool)
    order = np.zeros((S, T, n), np.int8)
    enc = np.zeros((S, T, W, H, 3), np.uint8)
    crc = np.zeros((S, T, n), np.uint32)
    crc_reset = np.zeros((S, n), np.uint32)
    crc_ctor = np.zeros((S, n), np.uint32)
    obs_full = np.zeros((F, T, n, P, P, 3), np.uint8)
    obs_reset_full = np.zeros((F, n, P, P, 3), np.uint8)
    mt_final = np.zeros((S, 624), np.uint32)
    mt_final_pos = np.zeros(S, np.int32)
    # 'rich' agents (base.py:461-471): the non-image fields, exactly as the reference hands them out
    # (NaN where the agent does not observe that field); index 0 on the time axis = reset(), t+1 = step t
    is_rich = any("rich" in a for a in spec["agents"])
    rich_reward = np.full((S, T + 1, n), np.nan)
    rich_position = np.full((S, T + 1, n, 2), np.nan)
    rich_orientation = np.full((S, T + 1, n), np.nan)

    def pov(o):
        return [x["pov"] if isinstance(x, dict) else x for x in o]

    def note_rich(si, ti, o):
        for k, x in enumerate(o):
            if isinstance(x, dict):
                assert set(x) <= {"pov", "reward", "position", "orientation"}
                if "reward" in x:
                    rich_reward[si, ti, k] = x["reward"]
                if "position" in x:
                    assert x["position"].dtype == np.float64
                    rich_position[si, ti, k] = x["position"]
                if "orientation" in x:
                    rich_orientation[si, ti, k] = x["orientation"]

    for si, seed in enumerate(seeds):
        env = refstate.make_ref_env(spec, recipe, seed=int(seed))
        c = refstate.canonical(env)
        for k in CANON:
            ctor[k].append(c[k])
        o = pov(env.gen_obs())
        crc_ctor[si] = [refstate.crc(x) for x in o]
        o = env.reset()
        note_rich(si, 0, o)
        o = pov(o)
        c = refstate.canonical(env)
        for k in CANON:
            rst[k].append(c[k])
        crc_reset[si] = [refstate.crc(x) for x in o]
        if si < F:
            obs_reset_full[si] = np.stack(o)
        if si < Fh:
            for k in range(n):
                obs_reset_a[k][si] = o[k]
        spy = refstate.OrderSpy(env.np_random)
        env.np_random = spy
        per = {k: [] for k in CANON}
        for t in range(T):
            o, r, dn, _ = env.step(actions[si, t])
            note_rich(si, t + 1, o)
            o = pov(o)
            assert all(x.min() >= 0 and x.max() <= 255 for x in o)
            c = refstate.canonical(env)
            for k in CANON:
                per[k].append(c[k])
            rewards[si, t], ep_done[si, t] = r, dn
            order[si, t] = spy.last
            enc[si, t] = env.grid.encode()
            crc[si, t] = [refstate.crc(x) for x in o]
            if si < F:
                obs_full[si, t] = np.stack(o)
            if si < Fh:
                for k in range(n):
                    obs_a[k][si, t] = o[k]
            if dn:
                env.reset()          # caller-side reset after done (README loop)
                reset_after[si, t] = True
        for k in CANON:
            rec[k].append(np.stack(per[k]))
        st = spy._rng.get_state()
        mt_final[si], mt_final_pos[si] = st[1], st[2]
    for k in CANON:
        d["step_" + k] = np.stack(rec[k])
        d["ctor_" + k] = np.stack(ctor[k])
        d["reset_" + k] = np.stack(rst[k])
    d.update(rewards=rewards, ep_done=ep_done, reset_after=reset_after, order=order, encode=enc,
             obs_crc=crc, obs_crc_reset=crc_reset, obs_crc_ctor=crc_ctor, obs_full=obs_full,
             obs_reset_full=obs_reset_full, mt_final=mt_final, mt_final_pos=mt_final_pos)
    if is_rich:
        d.update(rich_reward=rich_reward, rich_position=rich_position, rich_orientation=rich_orientation)
    if hetero:
        for k in range(n):
            d["obs_a%d" % k], d["obs_reset_a%d" % k] = obs_a[k], obs_reset_a[k]
    np.savez_compressed(out, **d)


FRAMES = {  # scenario -> (seeds, steps at which env.render() is captured)
    "MarlGrid-3AgentCluttered15x15-v0": ([1337, 1338], [0, 7, 31]),
    "Test-4AgentEmpty5x5-crowded": ([1337, 1340], [0, 5, 40]),
    "Goalcycle-demo-solo-v0": ([1337], [0, 12]),
    "Test-2AgentEmpty7x7-see-through": ([1337], [0, 9]),
    "Test-2AgentGoalcycle9x9-prestige": ([1337, 1339], [0, 25, 50]),
    "Test-3AgentEmpty7x11-nonsquare": ([1337], [0, 6]),
    "Test-3AgentCluttered12x6-nonsquare": ([1338], [3]),
}


def gen_frames(out):
    """Full-frame `MultiGridEnv.render(mode='rgb_array')` images (base.py:714-795): default arguments
    (highlight + agent-view side panels) and the bare grid (highlight=False, show_agent_views=False)."""
    d = {}
    for name, (seeds, steps) in FRAMES.items():
        spec = scenarios.registered(name)
        recipe = scenarios.ref_recipe(name)
        n = len(spec["agents"])
        for seed in seeds:
            env = refstate.make_ref_env(spec, recipe, seed=int(seed))
            env.reset()
            arng = np.random.RandomState(seed)
            acts = arng.choice(3, size=(max(steps) + 1, n)).astype(np.int8)
            d["%s/%d/actions" % (name, seed)] = acts
            for t in range(max(steps) + 1):
                if t in steps:
                    try:
                        full = env.render(mode="rgb_array")
                    except ValueError:
                        # upstream's side-panel arithmetic (base.py:766-783 mixes shape[0] / shape[1])
                        # breaks on non-square grids: capture the highlighted grid alone instead
                        full = env.render(mode="rgb_array", show_agent_views=False)
                        d["%s/%d/%d/nopanels" % (name, seed, t)] = np.array(1)
                    bare = env.render(mode="rgb_array", highlight=False, show_agent_views=False)
                    d["%s/%d/%d/full" % (name, seed, t)] = np.asarray(full).astype(np.uint8)
                    d["%s/%d/%d/bare" % (name, seed, t)] = np.asarray(bare).astype(np.uint8)
                env.step(acts[t])
    np.savez_compressed(out, **d)


FRAMES_BIG = {  # scenario -> (seed, steps, tile_size): whole-grid images too large to keep — their CRC32 and two crops are kept
    "Limit-3AgentSpawnRect150x150-prestige": (1337, [0, 6, 30], 8),      # 1 200 x 1 200 px; 'prestige' sprites in the corner
    "Limit-3AgentCluttered200x200-hide": (1337, [0, 9], 4),             # 800 x 800 px
    "Limit-2AgentEmpty255x255-view9-ts5": (1338, [0, 5], 8),             # 2 040 x 2 040 px: the largest grid
}


def gen_frames_big(out):
    """`MultiGridEnv.render(mode='rgb_array', tile_size=…, show_agent_views=False)` of grids beyond ~180 x 180 (base.py:714-759):
    highlighted and bare; per image its shape, the CRC32 of its bytes, the crop around the first agent and the top-left corner."""
    d = {}
    for name, (seed, steps, ts) in FRAMES_BIG.items():
        spec = scenarios.registered(name)
        recipe = scenarios.ref_recipe(name)
        n = len(spec["agents"])
        env = refstate.make_ref_env(spec, recipe, seed=int(seed))
        env.reset()
        arng = np.random.RandomState(seed)
        acts = arng.choice(3, size=(max(steps) + 1, n)).astype(np.int8)
        d["%s/actions" % name] = acts
        d["%s/seed" % name] = np.array(seed)
        d["%s/tile_size" % name] = np.array(ts)
        for t in range(max(steps) + 1):
            if t in steps:
                for tag, kw in (("full", dict()), ("bare", dict(highlight=False))):
                    img = np.asarray(env.render(mode="rgb_array", tile_size=ts, show_agent_views=False, **kw)).astype(np.uint8)
                    x, y = (int(v) for v in env.agents[0].pos)
                    r0, c0 = max(0, y * ts - 48), max(0, x * ts - 48)
                    d["%s/%d/%s/shape" % (name, t, tag)] = np.array(img.shape)
                    d["%s/%d/%s/crc" % (name, t, tag)] = np.array(refstate.crc(img), dtype=np.uint32)
                    d["%s/%d/%s/at" % (name, t, tag)] = np.array([r0, c0])
                    d["%s/%d/%s/near_agent0" % (name, t, tag)] = img[r0:r0 + 96, c0:c0 + 96].copy()
                    d["%s/%d/%s/corner" % (name, t, tag)] = img[:64, :64].copy()
            env.step(acts[t])
    np.savez_compressed(out, **d)


def gen_interact(m, out):
    """Hand-built pickup/drop/toggle scenes (base.py:587-617 — 'TODO: verify' in the reference).
    Each scene: EmptyMultiGrid 7x7, 2 agents teleported via a fresh grid + put_obj; scripted
    actions for agent 0 (agent 1 idles with 'done'); records canonical state + encode + obs CRC
    per step, or the exception name raised."""
    from marlgrid.objects import Box, Door, Key
    spec = scenarios.interact_spec()
    recipe = ("EmptyMultiGrid", dict(grid_size=7))
    scenes = scenarios.interact_scenes()
    d = {}
    for sname, sc in scenes.items():
        env = refstate.make_ref_env(spec, recipe, seed=1337)
        env.reset()
        # rebuild a deterministic layout: fresh walls+goal grid (EmptyMultiGrid._gen_grid draws
        # no random numbers), agents at fixed cells facing fixed dirs
        env._gen_grid(env.width, env.height)
        for k, (x, y, dr) in enumerate(sc["agents"]):
            a = env.agents[k]
            a.agents = []
            a.pos = None
            assert env.try_place_obj(a, np.array([x, y]))
            a.dir = dr
        mk = {"Box": lambda c, s: Box(c), "Door": lambda c, s: Door(color=c, state=s), "Key": lambda c, s: Key(c)}
        for (oid, x, y) in sc["objects"]:
            o = spec["objects"][oid]
            env.put_obj(None if o is None else mk[o["type"]](o["color"], o.get("state", 0)), x, y)
        for k, oid in sc.get("carrying", {}).items():
            o = spec["objects"][oid]
            env.agents[k].carrying = mk[o["type"]](o["color"], o.get("state", 0))
        per = {k: [] for k in CANON}
        encs, crcs, errs, rews = [], [], [], []
        for t, act in enumerate(sc["actions"]):
            try:
                o, r, dn, _ = env.step(act)
                err = ""
            except Exception as ex:       # noqa: BLE001 — the exception type is the golden
                err = type(ex).__name__
                o, r = None, np.zeros(len(env.agents))
            errs.append(err)
            c = refstate.canonical(env)
            for k in CANON:
                per[k].append(c[k])
            encs.append(env.grid.encode())
            rews.append(np.asarray(r, np.float64))
            crcs.append([refstate.crc(x) for x in o] if o is not None else [0] * len(env.agents))
        for k in CANON:
            d["%s/step_%s" % (sname, k)] = np.stack(per[k])
        d["%s/encode" % sname] = np.stack(encs)
        d["%s/obs_crc" % sname] = np.array(crcs, np.uint32)
        d["%s/error" % sname] = np.array(errs)
        d["%s/rewards" % sname] = np.stack(rews)
    np.savez_compressed(out, **d)


def main():
    m = refload.load()
    which = sys.argv[1:] or ["atlas", "rng", "occlusion", "traj", "interact", "frames"]
    if "atlas" in which:
        gen_atlas(m, os.path.join(HERE, "atlas.npz"))
    if "rng" in which:
        gen_rng(os.path.join(HERE, "rng.npz"))
    if "occlusion" in which:
        gen_occlusion(m, os.path.join(HERE, "occlusion.npz"))
    only = [w for w in which if w in TRAJ]
    for name in only:
        gen_traj(name, os.path.join(HERE, "traj_%s.npz" % name))
        print("traj", name, flush=True)
    if "traj" in which:
        for name in TRAJ:
            gen_traj(name, os.path.join(HERE, "traj_%s.npz" % name))
            print("traj", name, flush=True)
    if "frames" in which:
        gen_frames(os.path.join(HERE, "frames.npz"))
    if "frames" in which or "frames_big" in which:
        gen_frames_big(os.path.join(HERE, "frames_big.npz"))
    if "interact" in which:
        gen_interact(m, os.path.join(HERE, "interact.npz"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-60s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
