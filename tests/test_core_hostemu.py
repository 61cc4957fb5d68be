"""The engine's lane-per-env bodies (marlgrid_amd/csrc/mg_core.h — the text every lane of the HIP
seed / reset / step / place kernels runs) compiled for the host and stepped against the oracle.

What this pins without a GPU: the flat state machine (grid + packed agent records + stack ranks),
the fused auto-reset, the RNG stream (lazy MT19937 + look-ahead head, incl. the refill and the
conversion back to numpy's form) and rewards / done.  What it cannot pin: anything about the kernels
around the bodies (LDS staging, launch shapes) and the whole observation raster — those are the
`-m gpu` tests.  tests/native is test infrastructure; the product never loads it.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "native"))

import canon  # noqa: E402
import scenarios  # noqa: E402
from marlgrid_amd import seeding  # noqa: E402
from oracle import oracle as O  # noqa: E402

REW_TOL = 1e-6


def _same_state(emu, orc, what):
    st = emu.canonical()
    for b in range(emu.B):
        canon.assert_same(st[b], canon.oracle_canonical(orc.envs[b]), "%s env %d" % (what, b))
        assert seeding.same_stream(emu.numpy_rng_state(b), orc.envs[b].mt_state()), "%s env %d: RNG" % (what, b)


def test_numpy_form_round_trip():
    """lazy + head form -> numpy's (key, pos) for every phase of the block, against numpy itself"""
    import hostemu
    L = hostemu.lib()
    keys, lens = seeding.batch_keys([1337])
    mt = np.zeros((1, 624), np.uint32)
    pos = np.zeros(1, np.int32)
    head = np.zeros((1, 16), np.uint32)
    L.emu_mt_seed(1, hostemu._ptr(keys), hostemu._ptr(lens), hostemu._ptr(mt), hostemu._ptr(pos), hostemu._ptr(head))
    rs = np.random.RandomState()
    key = np.array(seeding.seed_words(1337), np.uint32)
    rs.seed(key)
    want = rs.get_state()
    got = seeding.numpy_form(mt[0], pos[0], 16)
    assert got[1] == want[2] == 624 and np.array_equal(got[0], want[1])
    # the head is the stream's first 16 outputs
    assert np.array_equal(head[0], rs.randint(0, 2 ** 32, size=16, dtype=np.uint64).astype(np.uint32))


@pytest.mark.parametrize("name,B,T,auto", [
    ("MarlGrid-3AgentCluttered11x11-v0", 48, 130, False),
    ("MarlGrid-3AgentCluttered15x15-v0", 32, 260, True),       # bench workload, auto-reset fused in step
    ("MarlGrid-4AgentEmpty9x9-v0", 40, 120, True),
    ("Custom-8AgentCluttered30x30", 8, 80, True),
    ("Test-4AgentEmpty5x5-crowded", 64, 100, True),
    ("Goalcycle-demo-solo-v0", 24, 300, True),
    ("Test-3AgentCluttered9x9-respawn", 48, 200, True),
    ("Test-4AgentEmpty5x5-respawn-noghost", 48, 150, False),
    ("Test-3AgentEmpty7x7-spawn-delay", 48, 90, True),
    ("Edge-16AgentEmpty6x6-view7", 16, 60, True),
    ("Edge-12AgentCluttered9x9-view3", 16, 80, False),
    ("Test-3AgentCluttered9x9-prestige-mixed", 32, 200, True),
    ("Test-2AgentRegion9x9", 32, 60, True),
    ("Test-3AgentSpawnRect9x9", 48, 200, True),      # agent_spawn_kwargs: reset, late spawn and respawn
    ("Test-2AgentReject9x9", 48, 200, True),         # place_obj(reject_fn=) tables in _gen_grid and agent_spawn_kwargs
    ("Test-3AgentSpawnRect9x9", 16, 130, False),
    ("Test-2AgentLateStatic10x10", 48, 160, True),   # static edits after random placements (fill ops in the reset program)
    ("Test-2AgentLateStatic10x10", 16, 110, False),
    ("Test-3AgentCluttered9x9-view6", 16, 70, True),
])
@pytest.mark.parametrize("par", [False, True])
def test_core_bodies_vs_oracle(name, B, T, auto, par):
    """par: the step as the obs kernel's fused step runs it — batches of 8 envs, the agents resolved by one lane per
    (agent, env) on the pre-step state (step_par_*), the sequential loop only for envs that ask for it"""
    import hostemu
    seeds = 4200 + np.arange(B)
    emu = hostemu.HostEmu(name, B, seeds, auto_reset=auto, par=par)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    _same_state(emu, orc, "%s ctor" % name)
    emu.reset()
    orc.reset()
    _same_state(emu, orc, "%s reset" % name)
    rng = np.random.RandomState(5)
    n = emu.n
    for t in range(T):
        a = rng.randint(0, 7, size=(B, n))
        r, d = emu.step(a)
        _o, r2, d2, _ = orc.step(a, render=False, auto_reset=auto)
        what = "%s step %d" % (name, t)
        assert np.abs(r.astype(np.float64) - r2).max() <= REW_TOL, what
        assert np.array_equal(d, d2), what
        if not auto and d.any():
            emu.reset(env_mask=d)
            for b in np.nonzero(d)[0]:
                orc.envs[b].reset()
        if t % 7 == 0 or t == T - 1:
            _same_state(emu, orc, what)
    assert not emu.error.any()


@pytest.mark.parametrize("name,B,T", [("MarlGrid-3AgentCluttered15x15-v0", 16, 6000),
                                       ("Custom-8AgentCluttered30x30", 4, 3000),
                                       ("Test-3AgentCluttered9x9-respawn", 16, 6000)])
@pytest.mark.parametrize("par", [False, True])
def test_core_bodies_long_horizon(name, B, T, par):
    """what bench.py times is ~60 000 steps per env: the lane-per-env bodies over MANY episodes (auto-reset inside the
    step, >= 25 wraps of the 624-word MT19937 state through the lazy regeneration) against the oracle — rewards / done
    every step, canonical state + RNG every 250 steps.  The GPU twin (LDS staging, LDS-DMA head refills, observations):
    tests/test_hip_soak.py."""
    import hostemu
    seeds = 515000 + np.arange(B)
    emu = hostemu.HostEmu(name, B, seeds, auto_reset=True, par=par)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    emu.reset()
    orc.reset()
    rng = np.random.RandomState(17)
    episodes = np.zeros(B, np.int64)
    for t in range(1, T + 1):
        a = rng.randint(0, 7, size=(B, emu.n))
        r, d = emu.step(a)
        _o, r2, d2, _ = orc.step(a, render=False, auto_reset=True)
        assert np.abs(r.astype(np.float64) - r2).max() <= REW_TOL and np.array_equal(d, d2), (name, t)
        episodes += d2
        if t % 250 == 0 or t == T:
            _same_state(emu, orc, "%s step %d" % (name, t))
    assert episodes.min() >= T // 100 and not emu.error.any()
    if par and emu.n <= 8:      # ghost_mode, nothing to pick up or toggle: no env-step of these scenarios needs the sequential loop
        assert emu.n_serial.value == 0, emu.n_serial.value


@pytest.mark.parametrize("name,B,T", [("Test-3AgentCluttered11x11-noghost", 64, 400), ("Test-4AgentEmpty5x5-crowded-noghost", 64, 300),
                                       ("Test-4AgentEmpty5x5-respawn-noghost", 64, 300), ("Test-4AgentEmpty5x5-ghost0", 64, 200),
                                       ("Test-4AgentEmpty5x5-crowded", 64, 300), ("Fuzz-3", 40, 120), ("Fuzz-8", 40, 120),
                                       ("Fuzz-21", 40, 120), ("Fuzz-27", 40, 120), ("Fuzz-33", 40, 120)])
def test_parallel_resolution_conflicts(name, B, T):
    """the agent-parallel resolution where it has to give way: without ghost_mode agents block each other (a forward move
    onto a cell somebody occupies or also heads for goes to the sequential loop), crowded rooms stack agents (ranks by turn
    order) — some env-steps take the loop, most do not, and every one equals the oracle"""
    import hostemu
    seeds = 90210 + np.arange(B)
    emu = hostemu.HostEmu(name, B, seeds, auto_reset=True, par=True)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    emu.reset()
    orc.reset()
    rng = np.random.RandomState(23)
    for t in range(T):
        a = rng.choice(7, size=(B, emu.n), p=[.15, .15, .5, .05, .05, .05, .05])
        r, d = emu.step(a)
        _o, r2, d2, _ = orc.step(a, render=False, auto_reset=True)
        assert np.abs(r.astype(np.float64) - r2).max() <= REW_TOL and np.array_equal(d, d2), (name, t)
        if t % 5 == 0 or t == T - 1:
            _same_state(emu, orc, "%s step %d" % (name, t))
    assert not emu.error.any()
    if emu.n <= 8:
        frac = emu.n_serial.value / float(B * T)
        assert frac < 0.9, frac                      # (the loop is the exception even in a crowded 5x5 room)
        if "noghost" in name:                        # (ghost_mode=0 `is not False`: moves may enter occupied cells, base.py:541)
            assert emu.n_serial.value > 0            # ... but it is taken


@pytest.mark.parametrize("par", [False, True])
@pytest.mark.parametrize("name,B,T", [("Limit-24AgentEmpty20x20-view5", 12, 150), ("Limit-3Agent100Kinds24x24", 12, 200),
                                       ("Limit-2Agent60Groups16x16", 12, 200)])
def test_core_bodies_beyond_the_old_limits(name, B, T, par):
    """24 agents (iter_order in the step's scratch column instead of sixteen nibbles), 115 object kinds (ids beyond 64), a reset
    program of 61 ops in "device" memory — the scenarios whose reference trajectories are tests/golden/traj_Limit-*.npz —
    against the oracle, Boxes picked up and dropped on the way (no toggles: a Box's raises TypeError upstream)"""
    import hostemu
    seeds = 77100 + np.arange(B)
    emu = hostemu.HostEmu(name, B, seeds, auto_reset=True, par=par)
    orc = O.OracleBatch(scenarios.registered(name), seeds)
    _same_state(emu, orc, "%s ctor" % name)
    emu.reset()
    orc.reset()
    rng = np.random.RandomState(29)
    for t in range(T):
        a = rng.choice(7, size=(B, emu.n), p=[.15, .15, .45, .1, .1, 0., .05])
        r, d = emu.step(a)
        _o, r2, d2, _ = orc.step(a, render=False, auto_reset=True)
        assert np.abs(r.astype(np.float64) - r2).max() <= REW_TOL and np.array_equal(d, d2), (name, t)
        if t % 10 == 0 or t == T - 1:
            _same_state(emu, orc, "%s step %d" % (name, t))
    assert not emu.error.any()
