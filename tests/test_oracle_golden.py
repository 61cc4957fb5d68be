"""CPU: the oracle (oracle/mg_oracle.c) replayed against the committed golden vectors that were
captured from the real reference (tests/golden/make_golden.py).  This is the oracle's pin."""
import os

import numpy as np
import pytest

import canon
import scenarios
from golden import refstate  # only for crc(); does not import the reference
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRAJ = sorted(f[5:-4] for f in os.listdir(GOLD) if f.startswith("traj_"))


def _canon_at(g, prefix, si, t=None):
    out = {}
    for k in canon.KEYS[:-1]:
        a = g[prefix + k][si]
        out[k] = a if t is None else a[t]
    return out


def _cmp_rich(orc, g, si, ti, what):
    """the oracle's 'rich' fields == what the reference handed out (exactly: float64 divides); NaN in the
    fixture = the agent does not observe that field"""
    for k in range(orc.n):
        r = orc.rich_obs(k)
        for key, want in (("reward", g["rich_reward"][si, ti, k]), ("orientation", g["rich_orientation"][si, ti, k])):
            if not np.isnan(want):
                assert r[key] == want, (what, k, key)
        want = g["rich_position"][si, ti, k]
        if not np.isnan(want).any():
            assert np.array_equal(r["position"], want), (what, k, r["position"], want)


def _cmp(orc, gold, what):
    c = canon.oracle_canonical(orc)
    for k in canon.KEYS[:-1]:
        assert np.array_equal(np.asarray(c[k]), np.asarray(gold[k])), "%s: %s\noracle=%r\ngolden=%r" % (
            what, k, c[k], gold[k])


@pytest.mark.parametrize("name", TRAJ)
def test_trajectory(name):
    g = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    spec = scenarios.registered(name)
    S, T, n = g["actions"].shape
    F = g["obs_full"].shape[0]
    for si in range(S):
        orc = O.make_env(spec, seed=int(g["seeds"][si]))
        _cmp(orc, _canon_at(g, "ctor_", si), "%s seed %d ctor" % (name, si))
        assert [refstate.crc(x) for x in orc.gen_obs()] == list(g["obs_crc_ctor"][si])
        o = orc.reset()
        _cmp(orc, _canon_at(g, "reset_", si), "%s seed %d reset" % (name, si))
        assert [refstate.crc(x) for x in o] == list(g["obs_crc_reset"][si])
        if si < F:
            assert np.array_equal(o, g["obs_reset_full"][si])
        Fh = g["obs_a0"].shape[0] if "obs_a0" in g.files else 0      # agents with their own views: per-agent arrays
        if si < Fh:
            assert all(np.array_equal(o[k], g["obs_reset_a%d" % k][si]) for k in range(n))
        rich = "rich_position" in g.files
        if rich:
            _cmp_rich(orc, g, si, 0, "%s seed %d reset" % (name, si))
        for t in range(T):
            o, r, dn, _, order = orc.step(g["actions"][si, t], return_order=True)
            what = "%s seed %d step %d" % (name, si, t)
            assert np.array_equal(order, g["order"][si, t]), what
            _cmp(orc, _canon_at(g, "step_", si, t), what)
            assert np.array_equal(r, g["rewards"][si, t]), what      # float64, bit-exact
            assert dn == g["ep_done"][si, t], what
            assert np.array_equal(orc.encode(), g["encode"][si, t]), what
            assert [refstate.crc(x) for x in o] == list(g["obs_crc"][si, t]), what
            if si < F:
                assert np.array_equal(o, g["obs_full"][si, t]), what
            if si < Fh:
                assert all(np.array_equal(o[k], g["obs_a%d" % k][si, t]) for k in range(n)), what
            if rich:
                _cmp_rich(orc, g, si, t + 1, what)
            if g["reset_after"][si, t]:
                orc.reset()
        mt, pos = orc.mt_state()
        assert pos == g["mt_final_pos"][si] and np.array_equal(mt, g["mt_final"][si]), name


def test_rng_seeding_and_draws():
    g = np.load(os.path.join(GOLD, "rng.npz"))
    L = O.lib()
    import ctypes as C
    for s, key in list(zip(g["seeds"], g["mt_key"])) + list(zip(g["special_seeds"], g["special_mt_key"])):
        k = O.seed_key(int(s))
        mt = np.zeros(624, np.uint32)
        pos = C.c_int32(0)
        L.mgo_mt_init_by_array(O._p(mt, C.c_uint32), C.byref(pos), O._p(k, C.c_uint32), len(k))
        assert pos.value == 624 and np.array_equal(mt, key), int(s)
    for i in range(4):
        mt = g["mt_key"][i].copy()
        pos = C.c_int32(624)
        raw = np.array([L.mgo_mt_next(O._p(mt, C.c_uint32), C.byref(pos)) for _ in range(2000)], np.uint32)
        assert np.array_equal(raw, g["raw_draws"][i])
        mt = g["mt_key"][i].copy()
        pos = C.c_int32(624)
        for t in range(300):
            x = L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), 14)
            y = L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), 10)
            assert (x, y) == tuple(g["randint_0_0_15_11"][i, t])
            a = list(range(8))
            for ii in range(7, 0, -1):
                j = L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), ii)
                a[ii], a[j] = a[j], a[ii]
            assert a == list(g["shuffle8"][i, t])


def test_rng_against_numpy_live():
    """numpy IS installed everywhere: pin the MT stream / bounded draws against it directly."""
    import ctypes as C
    L = O.lib()
    for seed in (0, 5, 1337, 2 ** 40 + 3):
        key = O.seed_key(seed)
        rs = np.random.RandomState()
        rs.seed([int(w) for w in key])
        mt = np.zeros(624, np.uint32)
        pos = C.c_int32(0)
        L.mgo_mt_init_by_array(O._p(mt, C.c_uint32), C.byref(pos), O._p(key, C.c_uint32), len(key))
        assert np.array_equal(rs.get_state()[1], mt)
        for t in range(500):
            hi = (int(rs.randint(1, 40)), int(rs.randint(1, 40)))   # consumes from rs ...
            L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), 38)   # ... mirror: randint(1,40) = 1+bounded(38)
            L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), 38)
            want = rs.randint((0, 0), hi)
            got = (L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), hi[0] - 1),
                   L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), hi[1] - 1))
            assert tuple(want) == got
            x = np.arange(5)
            rs.shuffle(x)
            a = list(range(5))
            for ii in range(4, 0, -1):
                j = L.mgo_bounded(O._p(mt, C.c_uint32), C.byref(pos), ii)
                a[ii], a[j] = a[j], a[ii]
            assert a == list(x)


def test_occlusion():
    g = np.load(os.path.join(GOLD, "occlusion.npz"))
    for key in g.files:
        if not key.startswith("T_"):
            continue
        vs, off = int(key.split("_")[1][2:]), int(key.split("_")[2][3:])
        N = 300
        T = np.unpackbits(g[key])[: N * vs * vs].reshape(N, vs, vs).astype(bool)
        M = np.unpackbits(g["M_" + key[2:]])[: N * vs * vs].reshape(N, vs, vs).astype(bool)
        for i in range(N):
            got = O.occlude(T[i], (vs // 2, vs - 1 - off))
            assert np.array_equal(got, M[i]), (key, i)


@pytest.mark.parametrize("ts", [5, 8, 11, 32])
def test_atlas(ts):
    g = np.load(os.path.join(GOLD, "atlas.npz"))
    colors = [str(c) for c in g["colors"]]
    spec = scenarios.interact_spec()
    spec["tile_size"] = ts
    spec["objects"] = spec["objects"] + [dict(type="BonusTile", color="yellow", state=0, reward=1)]
    for c0 in range(0, len(colors), 4):
        cs = colors[c0:c0 + 4]
        spec["agents"] = [dict(color=c) for c in cs]
        orc = O.OracleEnv(spec, construct=False)
        assert np.array_equal(orc.tile(0), g["empty_ts%d" % ts])
        assert np.array_equal(orc.tile(1), g["wall_ts%d" % ts])
        assert np.array_equal(orc.tile(2), g["goal_ts%d" % ts])
        assert np.array_equal(orc.tile(3), g["box_yellow_ts%d" % ts])
        assert np.array_equal(orc.tile(4), g["door_yellow_open_ts%d" % ts])
        assert np.array_equal(orc.tile(6), g["door_yellow_locked_ts%d" % ts])
        assert np.array_equal(orc.tile(9), g["bonus_ts%d" % ts])
        for k, c in enumerate(cs):
            for d in range(4):
                assert np.array_equal(orc.tile(0, k, d), g["agent_ts%d" % ts][c0 + k, d]), (c, d)
                assert np.array_equal(orc.tile(2, k, d), g["goal_blend_ts%d" % ts][c0 + k, d]), (c, d)


def _setup_scene(orc, sc):
    orc.regen_grid()
    for k, (x, y, d) in enumerate(sc["agents"]):
        orc.place_agent_at(k, x, y)
        orc.set_dir(k, d)
    for (oid, x, y) in sc["objects"]:
        orc.put_obj(oid, x, y)
    for k, oid in sc.get("carrying", {}).items():
        orc.set_carrying(k, oid)


@pytest.mark.parametrize("sname", sorted(scenarios.interact_scenes()))
def test_interact(sname):
    g = np.load(os.path.join(GOLD, "interact.npz"))
    spec = scenarios.interact_spec()
    sc = scenarios.interact_scenes()[sname]
    orc = O.OracleEnv(spec, seed=1337)
    orc.reset()
    _setup_scene(orc, sc)
    for t, act in enumerate(sc["actions"]):
        err = str(g["%s/error" % sname][t])
        what = "%s step %d" % (sname, t)
        if err in ("ValueError", "TypeError", "AssertionError", "AttributeError"):
            with pytest.raises({"ValueError": ValueError, "TypeError": TypeError, "AssertionError": AssertionError,
                                "AttributeError": AttributeError}[err]):
                orc.step(act)
            break      # state after a mid-loop exception depends on the shuffled order: not pinned
        o, r, dn, _ = orc.step(act)
        gold = {k: g["%s/step_%s" % (sname, k)][t] for k in canon.KEYS[:-1]}
        _cmp(orc, gold, what)
        assert np.array_equal(orc.encode(), g["%s/encode" % sname][t]), what
        if err == "":
            assert np.array_equal(r, g["%s/rewards" % sname][t]), what
            assert [refstate.crc(x) for x in o] == list(g["%s/obs_crc" % sname][t]), what
        else:
            assert err == "NameError"   # reference renderer crash on a closed door (objects.py:370)
