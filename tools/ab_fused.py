#!/usr/bin/env python
"""Interleaved A/B of the one-launch step (mg_step_render) between builds of the library given as paths (e.g. a
build of an earlier commit next to the current one).  First a parity check of every build against the first one
(two envs, same seeds and actions, 130 steps: observations, rewards, done, records, grids and the whole RNG state
must be equal), then 9 interleaved rounds of 100 launches each, all into the SAME observation buffer.
usage: [TILE=5 [VIEW=9] | PRESTIGE=3,8] [B=...] ab_fused.py path/to/ref.so path/to/new.so [...]"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

names = sys.argv[1:]
B = int(os.environ.get("B", "32768"))
WL = os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0")
g = torch.Generator().manual_seed(0)
PLACE = "search" if os.environ.get("PLACE", "1") != "0" else False


def build():
    if os.environ.get("TILE"):      # the bench scenario's shape with another view_tile_size (the instantiations off the fast path)
        from marlgrid_amd.agents import GridAgentInterface
        from marlgrid_amd.envs import ClutteredMultiGrid
        return ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=int(os.environ.get("VIEW", "7")), view_tile_size=int(os.environ["TILE"])) for c in ("red", "blue", "purple")],
                                  grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True, place_obs=PLACE)
    if os.environ.get("PRESTIGE"):  # PRESTIGE=<agents>,<tile>: the goal-cycle scenario of tools/bench_cases.py ('prestige'-coloured agents)
        from marlgrid_amd.agents import GridAgentInterface
        from marlgrid_amd.envs import ClutteredGoalCycleEnv
        k, ts = (int(x) for x in os.environ["PRESTIGE"].split(","))
        return ClutteredGoalCycleEnv(
            agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=ts, view_offset=1) for _ in range(k)],
            grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, reward_decay=False,
            initial_reward=True, penalty=-1.5, batch_size=B, strict=False, auto_reset=True, place_obs=PLACE)
    if os.environ.get("VIEW"):      # the bench scenario's shape with another view size at the registered 8-pixel tiles
        from marlgrid_amd.agents import GridAgentInterface
        from marlgrid_amd.envs import ClutteredMultiGrid
        return ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=int(os.environ["VIEW"]), view_tile_size=8) for c in ("red", "blue", "purple")],
                                  grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True, place_obs=PLACE)
    return make(WL, batch_size=B, auto_reset=True, strict=False, place_obs=PLACE)


env = build()
n = env.num_agents
_pm = getattr(env._groups[0], "placement_ms", None)
if _pm:
    print("obs buffers: kept %s of %d candidates (%s)" % (["%.4f" % c for c in _pm["kept"]], _pm["candidates"], _pm["stopped"]))
acts = [torch.randint(0, 7, (B, n), generator=g).cuda() for _ in range(16)]
vp, i32 = C.c_void_p, C.c_int32
libs = {}
for nm in names:
    L = C.CDLL(os.path.abspath(nm))
    L.mg_step_render.argtypes = [C.POINTER(N.Config), C.POINTER(N.State), vp, i32, vp, C.POINTER(N.GenProgram), vp, vp]
    L.mg_step_render.restype = i32
    L.mg_build_info.restype = C.c_char_p
    libs[nm] = L
    print("%s: %s" % (nm, L.mg_build_info().decode()))


RING = os.environ.get("RING", "0") != "0"      # RING=1: alternate between the env's two observation buffers, as env.step() does
                                              # (one buffer rewritten every launch partly lives in the 256 MiB Infinity Cache)


def launch(L, e, i):
    obs = e._ring[i % len(e._ring)]["obs"] if RING else e.obs
    rc = L.mg_step_render(C.byref(e._cfg), C.byref(e._state), acts[i % 16].data_ptr(), 8, e.rewards.data_ptr(),
                          C.byref(e._reset_prog), obs.data_ptr(), e._stream())
    assert rc == 0, rc


env.reset()
env.step(acts[0])          # (traces the reset program)
if os.environ.get("CHECK", "1") != "0":
    ref = build()
    ref.reset()
    ref.step(acts[0])
    for nm in names[1:]:
        for i in range(int(os.environ.get("STEPS", "130"))):
            launch(libs[names[0]], ref, i)
            launch(libs[nm], env, i)
            if i % 10 == 9 or 95 < i < 130 or i % 100 in (98, 99, 0, 1):
                for k in ("rewards", "done_t", "agent_state", "grid_state", "mt_state", "mt_pos", "mt_head", "step_count_t"):
                    assert torch.equal(getattr(env, k), getattr(ref, k)), ("%s differs from %s in %s at step %d" % (nm, names[0], k, i))
                for j in range(len(env._ring) if RING else 1):
                    k = "obs"
                    assert torch.equal(env._ring[j]["obs"] if RING else env.obs, ref._ring[j]["obs"] if RING else ref.obs), ("%s differs from %s in %s at step %d" % (nm, names[0], k, i))
        env.check_errors()
        print("%s identical to %s over 130 steps (obs, rewards, done, records, grids, RNG state)" % (nm, names[0]), flush=True)
    del ref
res = {nm: [] for nm in names}
for rep in range(int(os.environ.get("REPS", "9"))):
    for nm in names:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(libs[nm], env, 0)
        a.record()
        for i in range(100):
            launch(libs[nm], env, i)
        b.record()
        b.synchronize()
        res[nm].append(a.elapsed_time(b) / 100)
base = statistics.median(res[names[0]])
for nm in names:
    m = statistics.median(res[nm])
    print("%s median %.4f ms (min %.4f max %.4f)  %+.2f%% vs %s" % (nm, m, min(res[nm]), max(res[nm]), 100 * (m / base - 1), names[0]))
