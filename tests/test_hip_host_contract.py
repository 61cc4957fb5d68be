"""GPU tests of the host contract around the kernels (C ABI version 4): how per-env errors reach the caller
without a host sync per step, where the observation buffers come from, checkpoints."""
import ctypes as C

import numpy as np
import pytest

import product_envs

pytestmark = pytest.mark.gpu


def _bad_actions(B, n, env_b):
    import torch
    a = torch.zeros((B, n), dtype=torch.int64)
    a[env_b, 0] = 9                                   # ValueError upstream (base.py:619-620)
    return a


def test_strict_modes():
    """strict='sync' raises inside the step that caused the error (as upstream); strict=True (default) raises at
    the next call into the env once the GPU got there — without a synchronize on the step path —; strict=False
    only on check_errors().  All three leave the same state behind."""
    import torch
    B = 64
    envs = {m: product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=B, strict=m) for m in ("sync", True, False)}
    good = torch.zeros((B, 3), dtype=torch.int64)
    for env in envs.values():
        env.reset()
        env.step(good)
    with pytest.raises(ValueError, match="env 17"):
        envs["sync"].step(_bad_actions(B, 3, 17))
    e = envs[True]
    e.step(_bad_actions(B, 3, 17))                    # does not raise: nothing waited for the launch
    torch.cuda.synchronize()
    assert e._flag.raised()
    with pytest.raises(ValueError, match="env 17"):
        e.step(good)                                  # raised before anything is launched
    assert not e._flag.raised() and int(e.error_t.abs().sum()) == 0
    e.step(good)                                      # and the env keeps going
    f = envs[False]
    f.step(_bad_actions(B, 3, 17))
    f.step(good)
    with pytest.raises(ValueError, match="env 17"):
        f.check_errors()
    f.check_errors()                                  # cleared


def test_default_strict_step_does_not_synchronize():
    """env.step() with the defaults must not wait for the GPU: a step queued behind a long-running kernel
    returns while that kernel is still running."""
    import time
    import torch
    env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=4096)
    env.reset()
    a = torch.zeros((4096, 3), dtype=torch.int64, device=env.device)
    env.step(a)
    torch.cuda.synchronize()
    big = torch.empty(1 << 30, dtype=torch.uint8, device=env.device)
    t0 = time.perf_counter()
    for _ in range(40):
        big.fill_(1)                                  # ~6 ms of queued GPU work
    env.step(a)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert t_host < 0.5 * t_all, (t_host, t_all)


def test_obs_buffer_placement_search():
    """place_obs='search' (default): the observation ring is placed by the library (mg_obs_place: raw candidate
    allocations timed with the raster itself, bounded in memory and time); False = torch allocations.  Same observations
    either way; the candidates go back to the driver; a released ring of the fast class is remembered, so the second env
    of the same size in this process does not search; release_obs_cache() returns what is remembered."""
    import gc
    import torch
    from marlgrid_amd import _native as N
    from marlgrid_amd.base import _LibBuffer, release_obs_cache
    B = 16384                                          # 462 MB of observations per buffer: above the 256 MiB threshold
    # (what the process allocates once — code objects, the runtime's pools, a first env's launch — is not this test's
    # business: it has happened before `free0` is read, also when the test runs alone)
    warm = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=64)
    warm.reset()
    warm.step(torch.zeros((64, 3), dtype=torch.int64))
    torch.cuda.synchronize()
    del warm
    gc.collect()
    release_obs_cache()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    outs, found = {}, []
    for mode in ("search", False, "search"):
        env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=mode)
        env.reset()
        g = torch.Generator().manual_seed(3)
        for _ in range(5):
            o, r, d, _ = env.step(torch.randint(0, 7, (B, 3), generator=g))
        if mode == "search":
            pm = env._groups[0].placement_ms
            assert env.obs_placement == [pm]
            found.append(pm["found"])
            assert len(pm["kept"]) == 2 and pm["seconds"] < 8.0              # one pass of 2 s, allocations, the reused check
            assert pm["budget_bytes"] <= 32 << 30 or pm["reused"] == 2
            assert 2 * pm["buffer_bytes"] <= pm["pinned_bytes"] <= 2 * 3 * pm["buffer_bytes"]   # 1 .. 3 x per kept buffer
            if len(found) == 2 and found[0]:
                # the first env's ring was released when it died: taken back without a search
                assert pm["reused"] == 2 and pm["candidates"] == 0 and pm["found"] and pm["seconds"] < 0.25
            elif pm["reused"] == 0:
                assert 2 <= pm["candidates"] <= 194 and pm["passes"] == 1 and not pm["stirred"] and pm["block_pair_level"] == 0
                if pm["candidates"] <= N.PLACE_ALL:
                    assert sorted(pm["all"])[:2] == sorted(pm["kept"])      # the fastest two were kept
                if not pm["plain_stage"]:
                    # a candidate is a 2 P block followed by a P block, the buffer the window centred on their boundary
                    P = pm["candidate_bytes"] // 3
                    assert pm["candidate_bytes"] == 3 * P and P & (P - 1) == 0 and P >= pm["buffer_bytes"] / 2
                    assert abs(pm["window_offset"] + pm["buffer_bytes"] / 2 - 2 * P) <= 4096
            assert env.obs.data_ptr() % 4096 == 0
            assert torch.cuda.mem_get_info()[0] > free0 - (8 << 30)     # the rejected candidates are back
        else:
            assert env.obs_placement == []
        outs.setdefault(mode, []).append((o.cpu(), r.cpu(), d.cpu()))
        del env, o, r, d
        gc.collect()
    for got in outs["search"]:
        for x, y in zip(got, outs[False][0]):
            assert torch.equal(x, y)
    n_kept = release_obs_cache()
    assert n_kept == (2 if found[-1] else 0)
    # a library buffer by itself: usable by torch without a copy, freed with its last view
    mem = _LibBuffer(N.lib(), 100 << 20, torch.device("cuda", torch.cuda.current_device()))
    assert mem.ok
    t = mem.tensor((100 << 20,))
    assert t.data_ptr() == mem.ptr
    t.fill_(7)
    assert int(t[::4097].sum()) == 7 * len(t[::4097])
    del mem, t
    gc.collect()
    torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


def test_obs_buffer_placement_budgets_as_constructor_kwargs():
    """place_obs={...}: the search's memory and time budgets as the caller states them"""
    import gc
    env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=16384, place_obs={"budget": 8 << 30, "seconds": 1.0, "reuse": False})
    pm = env.obs_placement[0]
    assert pm["budget_bytes"] == 8 << 30 and pm["reused"] == 0 and pm["seconds"] < 4.0 and len(pm["kept"]) == 2
    assert pm["pinned_bytes"] <= 8 << 30
    with pytest.raises(ValueError):
        product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=64, place_obs={"megabytes": 5})
    del env
    gc.collect()


def test_obs_buffer_placement_search_when_nothing_is_found():
    """The search's later stages, forced (no candidate can be 60 % under the median; every allocation counts as slow),
    with place_obs='thorough' semantics: twelve misses, the one big allocate-and-free that stirs the driver's free lists,
    larger block pairs with several window positions measured per candidate, plain allocations, a second pass — then
    the best seen is kept, everything else goes back, and the env computes what an env on torch's buffers computes."""
    import gc
    import torch
    B = 16384
    env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=False)
    twin = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=False)
    env.reset()
    twin.reset()
    free0 = torch.cuda.mem_get_info()[0]
    env._place_obs_buffers(thorough=True, stir=True, reuse=False, gain=0.6, slow_alloc=-1.0, stir_cap=4 << 30, seconds=60.0,
                           max_candidates=40, budget=128 << 30, fast_rate=-1.0)
    pm = env._groups[0].placement_ms
    assert pm["found"] is False and pm["stopped"] == "cap" and pm["candidates"] == 82        # 2 + 2 passes of 40 (12 + 12 + 12 by level, 4 plain)
    assert pm["passes"] == 2                                         # nothing found: a second pass, from the best two of the first
    assert pm["stirred"] and pm["stirred"]["bytes"] == 4 << 30
    assert pm["block_pair_level"] == 2 and pm["plain_stage"] is True
    assert pm["windows_measured"] > pm["candidates"]                 # window positions besides the junction
    assert sorted(pm["all"])[:2] == sorted(pm["kept"])
    assert all(o % 4096 == 0 for o in pm["kept_window_offsets"])
    gc.collect()
    assert torch.cuda.mem_get_info()[0] > free0 - (8 << 30)          # the losers went back
    g = torch.Generator().manual_seed(11)
    for _ in range(4):
        a = torch.randint(0, 7, (B, 3), generator=g)
        for x, y in zip(env.step(a)[:3], twin.step(a)[:3]):
            assert torch.equal(x, y)


def test_obs_place_through_the_c_abi():
    """mg_obs_place / mg_obs_release / mg_obs_trim as a foreign binding calls them (ctypes, no package helper in
    between): buffers for the bench shard's launch config, 4 KiB-aligned, inside what the stats say they pin; the raster
    into them is what kept_ms says; released fast buffers come back without a search in <= 20 ms; a pointer that was not
    handed out is refused; MG_PLACE_NO_REUSE searches again; trim empties the record."""
    import ctypes as C
    import time
    import torch
    from marlgrid_amd import _native as N
    L = N.lib()
    B = 32768
    env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=False)
    env.reset()
    torch.cuda.synchronize()
    L.mg_obs_trim(-1)
    cfg, st = env._groups[0].cfg, env._state
    nbytes = B * 3 * 56 * 56 * 3
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def place(flags=0, budget=0, seconds=0.0):
        out, stats = (C.c_void_p * 2)(), N.PlaceStats()
        t0 = time.perf_counter()
        rc = L.mg_obs_place(C.byref(cfg), C.byref(st), 2, budget, seconds, flags, None, out, C.byref(stats), stream)
        return rc, [out[0], out[1]], stats, time.perf_counter() - t0

    free0 = torch.cuda.mem_get_info()[0]
    rc, bufs, s1, dt = place()
    assert rc == 0 and all(b and b % 4096 == 0 for b in bufs) and bufs[0] != bufs[1]
    assert s1.buffer_bytes == nbytes and s1.reused == 0 and s1.passes == 1 and s1.level == 0 and s1.stirred_bytes == 0
    assert s1.budget_bytes <= 32 << 30 and s1.seconds < 6.0 and 2 <= s1.candidates <= 194 and s1.windows >= s1.candidates
    assert 2 * nbytes <= s1.pinned_bytes <= 6 * nbytes == 6 * s1.buffer_bytes
    assert free0 - torch.cuda.mem_get_info()[0] <= s1.pinned_bytes + (64 << 20)          # every other candidate is back
    for i in range(2):
        assert s1.window_offset[i] + nbytes <= s1.arena_bytes[i]
    # the buffers are ordinary device memory: the raster into one of them is what the search measured
    ms = C.c_float(0)
    N.check(L.mg_time_render_obs(C.byref(cfg), C.byref(st), bufs[0], 5, C.byref(ms), stream))
    assert abs(ms.value - s1.kept_ms[0]) <= 0.12 * s1.kept_ms[0], (ms.value, s1.kept_ms[0])
    if s1.found:
        # (found: the kept pair is 12 % under the median candidate, or takes the raster's bytes at the fast class's 5.9 TB/s)
        assert s1.kept_ms[1] <= 0.88 * s1.median_ms * 1.001 or nbytes / (s1.kept_ms[1] * 1e-3) >= 5.9e12
    assert L.mg_obs_release(C.c_void_p(bufs[0] + 4096)) == -100                           # not a placed buffer
    for b in bufs:
        assert L.mg_obs_release(C.c_void_p(b)) == 0
    assert L.mg_obs_release(C.c_void_p(bufs[0])) == -100                                  # released twice
    if s1.found:
        rc, bufs2, s2, dt2 = place()
        assert rc == 0 and sorted(bufs2) == sorted(bufs) and s2.reused == 2 and s2.candidates == 0 and s2.found == 1
        assert dt2 <= 0.020, dt2                                                          # remembered: no search
        for b in bufs2:
            assert L.mg_obs_release(C.c_void_p(b)) == 0
        rc, bufs3, s3, _ = place(flags=N.PLACE_NO_REUSE, seconds=1.0)
        assert rc == 0 and s3.reused == 0 and s3.candidates >= 2
        for b in bufs3:
            assert L.mg_obs_release(C.c_void_p(b)) == 0
    assert L.mg_obs_trim(-1) >= (2 if s1.found else 0)
    assert L.mg_obs_trim(-1) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)
    # too little memory allowed for even one candidate: the plain buffers are still handed out (found = 0), nothing leaks
    rc, bufs4, s4, _ = place(budget=1 << 20, seconds=0.5)
    # (found = 1 only if the plain pair happens to take the raster's bytes at the fast class's 5.9 TB/s)
    assert rc == 0 and s4.pinned_bytes == 2 * nbytes
    assert (s4.found == 0 and s4.stopped in (2, 3, 4)) or (s4.found == 1 and nbytes / (max(s4.kept_ms[0], s4.kept_ms[1]) * 1e-3) >= 5.9e12)
    for b in bufs4:
        assert L.mg_obs_release(C.c_void_p(b)) == 0
    L.mg_obs_trim(-1)                                   # (a pair that counted as found is remembered when it is released)
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


def test_state_dict_round_trip_and_versioning():
    import torch
    env = product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=32, auto_reset=True)
    env.reset()
    g = torch.Generator().manual_seed(5)
    acts = [torch.randint(0, 7, (32, 3), generator=g) for _ in range(30)]
    for a in acts[:10]:
        env.step(a)
    sd = env.state_dict()
    assert int(sd["version"]) == 3
    ref = [tuple(x.clone() for x in env.step(a)[:3]) for a in acts[10:]]
    env2 = product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=32, auto_reset=True)
    env2.load_state_dict(sd)
    for a, (o, r, d) in zip(acts[10:], ref):
        o2, r2, d2, _ = env2.step(a)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
    old = {k: v for k, v in sd.items() if k not in ("version", "mt_head")}       # a round-1 checkpoint
    with pytest.raises(KeyError, match="mt_head"):
        env2.load_state_dict(old)
    with pytest.raises(ValueError, match="version"):
        env2.load_state_dict(dict(sd, version=torch.tensor(2)))


def test_c_abi_rejects_bad_viewer_subsets():
    """check_cfg: n_view / view_agent out of range are argument errors (they would index LDS out of range);
    mg_step_render takes no viewer subset."""
    from marlgrid_amd import _native as N
    env = product_envs.build("MarlGrid-3AgentCluttered11x11-v0", batch_size=8)
    env.reset()
    L = N.lib()
    cfg = N.Config.from_buffer_copy(env._cfg)
    cfg.n_view = 4                                     # > n_agents
    assert L.mg_render_obs(C.byref(cfg), C.byref(env._state), env.obs.data_ptr(), None, None, None, env._stream()) == -100
    cfg.n_view = 1
    cfg.view_agent[0] = 3                              # >= n_agents
    assert L.mg_render_obs(C.byref(cfg), C.byref(env._state), env.obs.data_ptr(), None, None, None, env._stream()) == -100
    cfg.view_agent[0] = 2
    import torch
    a = torch.zeros((8, 3), dtype=torch.int64, device=env.device)
    assert L.mg_step_render(C.byref(cfg), C.byref(env._state), a.data_ptr(), 8, env.rewards.data_ptr(), None,
                            env.obs.data_ptr(), env._stream()) == -100
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_shard_pipeline_equals_one_env():
    """ShardPipeline: the batch as two envs on two streams (overlapping launches) — the same trajectories as ONE env
    of the whole batch, env by env, and the parts really run on their own streams"""
    import torch
    from marlgrid_amd.envs import make
    from marlgrid_amd.sharding import ShardPipeline
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 512
    one = make(name, batch_size=B, seeds=[1337 + g for g in range(B)], auto_reset=True)
    pipe = ShardPipeline(lambda **kw: make(name, auto_reset=True, **kw), B, parts=2, seed=1337)
    assert pipe.streams[0].cuda_stream != pipe.streams[1].cuda_stream and pipe.part_size == 256
    o = one.reset()
    parts = pipe.reset()
    pipe.synchronize()
    for k in range(2):
        assert torch.equal(pipe.part(k, o), parts[k])
    rng = np.random.RandomState(5)
    for t in range(120):            # past max_steps: every env resets inside a launch at least once
        a = torch.from_numpy(rng.randint(0, 7, size=(B, 3))).cuda()
        o, r, d, _ = one.step(a)
        res = pipe.step(a)
        pipe.synchronize()
        for k in range(2):
            o2, r2, d2, _ = res[k]
            assert torch.equal(pipe.part(k, o), o2) and torch.equal(pipe.part(k, r), r2) and torch.equal(pipe.part(k, d), d2), (t, k)
    pipe.check_errors()
    with pytest.raises(ValueError):
        ShardPipeline(lambda **kw: None, 7, parts=2)


@pytest.mark.gpu
def test_pipeline_public_path_alternating_loop_equals_one_env():
    """make(id, pipeline=2) and MultiGridEnv.pipelined(learners, parts=2): the double-buffered loop — part after part,
    each under its own stream, policy included — yields, env by env, the trajectories of ONE env of the whole batch;
    an error recorded by a part surfaces through pipe.check_errors() from any stream."""
    import torch
    from marlgrid_amd.agents import IndependentLearners, LearningAgent
    from marlgrid_amd.envs import ClutteredMultiGrid, make
    name, B = "MarlGrid-3AgentCluttered11x11-v0", 512
    one = make(name, batch_size=B, seeds=[1337 + g for g in range(B)], auto_reset=True)
    pipe = make(name, pipeline=2, batch_size=B, seed=1337, auto_reset=True)
    assert pipe.parts == 2 and pipe.num_agents == 3 and len(pipe.agents) == 3
    o = one.reset()
    obs = pipe.reset()
    rng = np.random.RandomState(9)
    for t in range(110):
        a = torch.from_numpy(rng.randint(0, 7, size=(B, 3))).cuda()
        o, r, d, _ = one.step(a)
        for k in range(2):
            with pipe.on(k):
                assert torch.cuda.current_stream().cuda_stream == pipe.streams[k].cuda_stream
                o2, r2, d2, _ = pipe.step_part(k, pipe.part(k, a).contiguous())
        pipe.synchronize()
        for k in range(2):
            env = pipe.envs[k]
            assert torch.equal(pipe.part(k, o), env.obs) and torch.equal(pipe.part(k, r), env.rewards) \
                and torch.equal(pipe.part(k, d), env.done_b), (t, k)
    # an invalid action in part 1, stepped under its own stream; checked from the default stream
    bad = torch.full((B // 2, 3), 9, dtype=torch.int64, device="cuda")
    with pipe.on(1):
        pipe.envs[1].strict = False
        pipe.step_part(1, bad)
    with pytest.raises(ValueError):
        pipe.check_errors()

    class Rand(LearningAgent):
        seen = 0

        def action_step(self, obs):
            return torch.randint(0, 3, (obs.shape[0],), device=obs.device)

        def save_step(self, obs, act, nxt, rew, done):
            Rand.seen += obs.shape[0]

    learners = IndependentLearners(Rand(color="red", view_tile_size=8), Rand(color="blue", view_tile_size=8))
    pipe2 = ClutteredMultiGrid.pipelined(learners, parts=2, grid_size=9, n_clutter=4, batch_size=64, auto_reset=True)
    assert pipe2.envs[0].agents[0] is learners[0] and pipe2.envs[1].agents[0] is not learners[0]
    obs = pipe2.reset()
    with learners.episode():
        for t in range(12):
            for k in range(2):
                with pipe2.on(k):
                    act = learners.action_step(obs[k])
                    nxt, rew, done, _ = pipe2.step_part(k, act)
                    learners.save_step(obs[k], act, nxt, rew, done)
                    obs[k] = nxt
    pipe2.check_errors()
    assert Rand.seen == 12 * 2 * 32 * 2


def _encode_reference_torch(env):
    """MultiGrid.encode (base.py:196-214) for the whole batch in torch: the base object's triple; on an empty cell the triple
    of the lowest-rank placed agent standing there (agents written in descending rank order: the lowest rank lands last)"""
    import torch
    from marlgrid_amd import _native as N
    B, W, H, n = env.batch_size, env.width, env.height, env.num_agents
    table = torch.tensor([[0, 0, 0]] + [list(o.encode()) for o in env.obj_reg.objs[1:]], dtype=torch.uint8, device=env.device)
    base = env.grid_state[:, :W * H].long()
    enc = table[base]                                                     # (B, W*H, 3)
    rec = env.agent_state
    by = lambda i: (rec >> (8 * i)) & 0xFF                                 # noqa: E731
    x, y, d, fl, rk = by(N.AG_X), by(N.AG_Y), by(N.AG_DIR), by(N.AG_FLAGS), by(N.AG_RANK)
    cell = (x * H + y).clamp(max=W * H - 1)
    colors = torch.tensor([env._cfg.agent_color_idx[k] for k in range(n)], device=env.device)
    rows = torch.arange(B, device=env.device)
    for want_rank in range(n - 1, -1, -1):
        for k in range(n):
            m = ((fl[:, k] & N.AF_PLACED) != 0) & (rk[:, k] == want_rank) & (base[rows, cell[:, k]] == 0)
            tri = torch.stack([torch.full((B,), 13, device=env.device), colors[k].expand(B), d[:, k]], dim=1).to(torch.uint8)
            enc[rows[m], cell[m, k]] = tri[m]
    return enc.view(B, W, H, 3)


@pytest.mark.parametrize("name,B", [("MarlGrid-3AgentCluttered15x15-v0", 32768), ("MarlGrid-3AgentCluttered11x11-v0", 4096),
                                    ("MarlGrid-3AgentCluttered15x15-v0", 80001),      # 2 198 pieces: the workgroup-wide store phase
                                    ("Test-4AgentEmpty5x5-crowded", 1000), ("Custom-8AgentCluttered30x30", 4099),
                                    ("Edge-2AgentCluttered40x40-view9-off3", 700), ("Edge-16AgentEmpty6x6-view7", 333),
                                    ("MarlGrid-2AgentEmpty9x9-v0", 1)])
def test_encode_whole_batch_and_through_the_c_abi(name, B):
    """mg_encode (MultiGrid.encode, base.py:196-214) for the WHOLE batch against a torch restatement — the kernel works on
    pieces of the flat output stream that do not know about envs, so every env boundary, chunk phase and piece boundary is
    in here (4 096-cell pieces at the bench shard, 1 024-cell ones below) —, against the oracle on sampled envs, with a
    vis_mask, and into a caller's buffer that is NOT 16-byte aligned (pure C ABI: byte stores instead of 16-byte ones)."""
    import torch
    import scenarios
    from marlgrid_amd import _native as N
    from oracle import oracle as O
    seeds = 2024 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds, place_obs=False)
    env.reset()
    rng = np.random.RandomState(1)
    n = env.num_agents
    for t in range(12):
        env.step(torch.from_numpy(rng.randint(0, 3, size=(B, n))))
    got = env.grid.encode()
    want = _encode_reference_torch(env)
    assert torch.equal(got, want), torch.nonzero((got != want).any(dim=-1))[:5]
    ids = sorted({0, 1, B // 3, B // 2, B - 2, B - 1} & set(range(B)))
    orc = O.OracleBatch(scenarios.registered(name), seeds[ids])
    orc.reset()
    rng = np.random.RandomState(1)
    for t in range(12):
        orc.step(rng.randint(0, 3, size=(B, n))[ids], render=False)
    for j, b in enumerate(ids):
        assert np.array_equal(got[b].cpu().numpy(), orc.envs[j].encode()), b
    # vis_mask: (B, W, H) — cells that are not visible encode as zeros
    vm = torch.from_numpy(np.random.RandomState(2).rand(B, env.width, env.height) < 0.6).to(env.device)
    masked = env.grid.encode(vm)
    assert torch.equal(masked, want * vm[..., None].to(torch.uint8))
    # a misaligned output buffer through the C ABI
    nb = B * env.width * env.height * 3
    for shift in (1, 7):
        buf = torch.full((nb + 32,), 0x5A, dtype=torch.uint8, device=env.device)
        N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, C.c_void_p(buf.data_ptr() + shift), env._stream()))
        assert torch.equal(buf[shift:shift + nb].view_as(want), want)
        assert bool((buf[:shift] == 0x5A).all()) and bool((buf[shift + nb:] == 0x5A).all())


@pytest.mark.parametrize("name,B", [("MarlGrid-3AgentCluttered15x15-v0", 32768), ("MarlGrid-3AgentCluttered15x15-v0", 4099),
                                    ("MarlGrid-3AgentCluttered11x11-v0", 4096), ("Test-4AgentEmpty5x5-crowded", 1000),
                                    ("Custom-8AgentCluttered30x30", 2051), ("Edge-2AgentCluttered40x40-view9-off3", 700),
                                    ("Edge-16AgentEmpty6x6-view7", 333), ("Goalcycle-demo-solo-v0", 517),
                                    ("Test-3AgentCluttered9x9-prestige-mixed", 300), ("Limit-24AgentEmpty20x20-view5", 77),
                                    ("Limit-3Agent100Kinds24x24", 130), ("Limit-2AgentCluttered128x128", 9), ("Limit-3AgentCluttered200x200-hide", 5),
                                    ("Limit-2Agent60Groups16x16", 203), ("FuzzW-0", 21), ("MarlGrid-2AgentEmpty9x9-v0", 1)])
def test_encode_in_step(name, B):
    """encode_in_step=True: `env.grid_encoding` after every step() (and reset()) is what `env.grid.encode()` — the
    stand-alone mg_encode kernel, itself compared with the reference's encode at every step of every golden — returns at
    that moment, and a twin without the option steps identically (observations, rewards, done, state).  Where the step's
    own launch writes it (mg_step_render_encode: grids staged in LDS, object ids and agent marks in one byte) and where
    the library declines and the host launches mg_encode behind the step (a 200 x 200 grid read in place; agents with
    their own views), with auto-reset inside the launch, batches that end inside a wave's run of envs,
    16 x 16-cell grids (no padding between the staged envs) and the flat stream at every 16-byte phase."""
    import torch
    seeds = 77 + np.arange(B)
    env = product_envs.build(name, batch_size=B, seeds=seeds, place_obs=False, auto_reset=True, encode_in_step=True)
    twin = product_envs.build(name, batch_size=B, seeds=seeds, place_obs=False, auto_reset=True)
    o1, o2 = env.reset(), twin.reset()
    assert torch.equal(env.grid_encoding, env.grid.encode())
    rng = np.random.RandomState(3)
    n = env.num_agents
    T = 40 if B <= 4099 else 14
    for t in range(T):
        a = rng.randint(0, 7, size=(B, n))
        if name in ("Limit-3Agent100Kinds24x24", "Limit-2Agent60Groups16x16"):
            a[a == 5] = 6          # (toggling a Box is a TypeError upstream, objects.py: reproduced, and not this test's subject)
        a = torch.from_numpy(a)
        o1, r1, d1, _ = env.step(a)
        o2, r2, d2, _ = twin.step(a)
        want = env.grid.encode()
        assert torch.equal(env.grid_encoding, want), (t, torch.nonzero((env.grid_encoding != want).any(dim=-1))[:5])
        if t % 5 == 0 or t == T - 1:
            assert torch.equal(want, _encode_reference_torch(env)), t
            assert torch.equal(r1, r2) and torch.equal(d1, d2), t
            assert all(torch.equal(x, y) for x, y in zip(_obs_list(o1), _obs_list(o2))), t
    assert torch.equal(env.grid_state, twin.grid_state) and torch.equal(env.agent_state, twin.agent_state)
    assert torch.equal(env.mt_pos, twin.mt_pos)
    # who wrote it: the step's own launch for the shapes the library has the encode compiled into (views 7 / 9 / any at 8-pixel
    # tiles, view 7 at 5-pixel tiles; MG_RENDER_GROUP_N) — the host's mg_encode launch behind the step where it declined
    if env.fused_step and not env._hetero:
        in_launch = {"MarlGrid-3AgentCluttered15x15-v0": True, "MarlGrid-3AgentCluttered11x11-v0": True, "Custom-8AgentCluttered30x30": True,
                     "MarlGrid-2AgentEmpty9x9-v0": True, "Limit-3AgentCluttered200x200-hide": False,
                     "Test-3AgentCluttered9x9-prestige-mixed": False}
        if name in in_launch:
            assert env._enc_fused == in_launch[name], (name, env.kernel_name, env._enc_fused)
    env.check_errors()


def _obs_list(o):
    import torch
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [v for v in o.values() if torch.is_tensor(v)]
    out = []
    for x in o:
        out.extend(_obs_list(x))
    return out


def test_obs_ring_longer_than_a_placement_call():
    """obs_buffers > MG_PLACE_MAX (8): the first eight buffers of the ring are placed by the library, the rest stay torch
    allocations, the ring rotates through all of them (the constructor used to fail with 'invalid argument')"""
    import gc
    import torch
    from marlgrid_amd import _native as N
    B = 12288                                         # 347 MB per buffer: above the 256 MiB threshold
    env = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, obs_buffers=9)
    twin = product_envs.build("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, place_obs=False)
    pm = env.obs_placement[0]
    assert pm["placed_buffers"] == N.PLACE_MAX == 8 and pm["ring_buffers"] == 9 and len(pm["kept"]) == 8
    assert len({t.data_ptr() for t in env._groups[0].ring}) == 9
    env.reset(); twin.reset()
    g = torch.Generator().manual_seed(4)
    seen = set()
    for t in range(20):
        a = torch.randint(0, 7, (B, 3), generator=g)
        o, r, d, _ = env.step(a)
        o2, r2, d2, _ = twin.step(a)
        seen.add(o.data_ptr())
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
    assert len(seen) == 9
    del env, twin
    gc.collect()
