"""Build-container only (skipped where /root/reference is absent): the oracle stepped side by side
with the REAL reference (imported through tests/golden/refshim) on fresh seeds / actions that are
not in the committed fixtures."""
import numpy as np
import pytest

import canon
import refload
import scenarios
from oracle import oracle as O

pytestmark = pytest.mark.reference


@pytest.mark.parametrize("name,seed,T", [("MarlGrid-3AgentCluttered15x15-v0", 9001, 250),
                                          ("Test-4AgentEmpty5x5-crowded-noghost", 9002, 200),
                                          ("Goalcycle-demo-solo-v0", 9003, 200),
                                          ("Custom-8AgentCluttered30x30", 9004, 60)])
def test_live_side_by_side(name, seed, T):
    import refstate
    spec = scenarios.registered(name)
    env = refstate.make_ref_env(spec, scenarios.ref_recipe(name), seed=seed)
    orc = O.OracleEnv(spec, seed=seed)
    n = len(spec["agents"])
    rng = np.random.RandomState(seed)

    def same_state(what):
        a, b = refstate.canonical(env), canon.oracle_canonical(orc)
        for k in canon.KEYS[:-1]:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (what, k)

    same_state("ctor")
    assert np.array_equal(np.stack(env.reset()), orc.reset())
    same_state("reset")
    for t in range(T):
        a = rng.randint(0, 7, size=n)
        o1, r1, d1, _ = env.step(a)
        o2, r2, d2, _ = orc.step(a)
        assert np.array_equal(np.stack(o1), o2), t
        assert np.array_equal(r1, r2) and d1 == d2, t
        assert np.array_equal(env.grid.encode(), orc.encode()), t
        same_state("step %d" % t)
        if d1:
            assert np.array_equal(np.stack(env.reset()), orc.reset())
    st = env.np_random.get_state()
    mt, pos = orc.mt_state()
    assert st[2] == pos and np.array_equal(st[1], mt)


@pytest.mark.parametrize("i", list(range(40)) + [102, 115, 124, 167, 181, 202, 349, 371, 380, 399, 572])
def test_live_fuzz(i):
    """random constructor knobs (scenarios.fuzz_case): reference == oracle, every step"""
    test_live_side_by_side("Fuzz-%d" % i, 31000 + i, 90)


def test_occlusion_live_random():
    m = refload.load()
    from marlgrid.agents import occlude_mask
    rng = np.random.RandomState(123)
    for vs in (3, 5, 7, 9, 13):
        for off in (0, 1):
            for _ in range(150):
                T = rng.rand(vs, vs) < rng.choice([0.6, 0.8, 0.95])
                want = occlude_mask(T.copy(), (vs // 2, vs - 1 - off))
                assert np.array_equal(O.occlude(T, (vs // 2, vs - 1 - off)), want)


def test_live_place_obj_and_try_place_obj():
    """place_obj / try_place_obj on a live grid (base.py:664-708): positions, success flags and RNG
    consumption of the oracle against the reference's own methods."""
    import refstate
    from marlgrid.objects import Wall, Goal
    name = "MarlGrid-3AgentCluttered11x11-v0"
    spec = scenarios.registered(name)
    for seed in (1, 2, 3, 4):
        env = refstate.make_ref_env(spec, scenarios.ref_recipe(name), seed=seed)
        orc = O.OracleEnv(spec, seed=seed)
        env.reset(); orc.reset()
        for i in range(6):
            pos = env.place_obj(Wall(), max_tries=100)
            assert tuple(pos) == orc.place_obj(1, max_tries=100)
        pos = env.place_obj(Goal(color="green", reward=1), top=(2, 3), size=(4, 20), max_tries=100)
        assert tuple(pos) == orc.place_obj(2, region=(2, 3, 6, 11), max_tries=100)
        rng = np.random.RandomState(seed)
        for i in range(30):
            x, y = int(rng.randint(0, 11)), int(rng.randint(0, 11))
            ok = env.try_place_obj(Wall(), np.array([x, y]))
            assert bool(ok) == orc.try_place_obj(1, x, y), (seed, i, x, y)
        a, b = refstate.canonical(env), canon.oracle_canonical(orc)
        for k in canon.KEYS[:-1]:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        st = env.np_random.get_state()
        mt, p = orc.mt_state()
        assert st[2] == p and np.array_equal(st[1], mt)
