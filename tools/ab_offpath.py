#!/usr/bin/env python
"""The instantiations off the <7, 8, 16> fast path, each under the workgroup shapes it can run with (measurement
build: MG_RENDER_WPB read at every launch; "0" = the launcher's own choice): the reference's default tile 5, tile 6,
tile 11, three 'prestige' agents at tile 8, and the reference's one runnable example (examples/human_player.py:
goal cycle 13x13, ONE 'prestige' agent, view 7, tile 11, view_offset 1).  Per shape: the raster alone
(mg_render_obs) and the whole step (mg_step_render), interleaved rounds, fraction of 8 TB/s on the raster's bytes.
usage: ab_offpath.py [case substring ...]"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
N.use_library(os.path.join(ROOT, "marlgrid_amd", "csrc", os.environ.get("AB_LIB", "libmarlgrid_hip_ab.so")))   # the measurement build
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredGoalCycleEnv, ClutteredMultiGrid  # noqa: E402

COLS = ["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]
B = int(os.environ.get("B", "32768"))
# observation buffers chosen by the product's placement search (default), or PLACE=0: plain torch allocations, whose
# class varies from buffer to buffer by up to 25 % (what the tile >= 8 cases are sensitive to)
PLACE = False if os.environ.get("PLACE", "1") == "0" else "search"


def agents(n, vs, ts, **kw):
    return [GridAgentInterface(color=COLS[k % 8], view_size=vs, view_tile_size=ts, **kw) for k in range(n)]


def cluttered(ts):
    return lambda: ClutteredMultiGrid(agents=agents(3, 7, ts), grid_size=15, clutter_density=0.15, batch_size=B,
                                      strict=False, auto_reset=True, place_obs=PLACE)


def goalcycle(n, ts):
    return lambda: ClutteredGoalCycleEnv(
        agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=ts, view_offset=1) for _ in range(n)],
        grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, reward_decay=False,
        initial_reward=True, penalty=-1.5, batch_size=B, strict=False, auto_reset=True, place_obs=PLACE)


# shapes: "<waves per workgroup>" (0 = the launcher's choice), "+rt" = the run-time-tile-size instantiation
cases = [("tile 5 (GridAgentInterface default)", cluttered(5), ["0", "0+rt"]),
         ("tile 6", cluttered(6), ["0", "0+rt"]),
         ("tile 11", cluttered(11), ["0", "0+rt"]),
         ("goal cycle, 3 'prestige' agents, tile 8", goalcycle(3, 8), ["0", "8"]),
         ("human_player: goal cycle, 1 'prestige' agent, tile 11", goalcycle(1, 11), ["0", "0+rt", "8"])]
only = sys.argv[1:]
for label, mk, shapes in cases:
    if only and not any(o in label for o in only):
        continue
    env = mk()
    env.reset()
    L = env._lib
    g = torch.Generator().manual_seed(0)
    acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).to(env.device) for _ in range(8)]
    env.step(acts[0])
    ms = C.c_float(0)
    res = {w: {"raster": [], "step": []} for w in shapes}
    for rep in range(5):
        for w in shapes:
            os.environ["MG_RENDER_WPB"] = w.split("+")[0]
            os.environ["MG_RENDER_RT_TS"] = "1" if w.endswith("+rt") else "0"
            N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 20, C.byref(ms), env._stream()))
            res[w]["raster"].append(ms.value)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            env.step(acts[0])
            a.record()
            for i in range(20):
                env.step(acts[i % 8])
            b.record()
            b.synchronize()
            res[w]["step"].append(a.elapsed_time(b) / 20)
    os.environ["MG_RENDER_WPB"] = "0"
    os.environ["MG_RENDER_RT_TS"] = "0"
    env.check_errors()
    nb = env.obs.numel()
    for w in shapes:
        r, s = statistics.median(res[w]["raster"]), statistics.median(res[w]["step"])
        print(json.dumps({"case": label, "shape": {"0": "launcher's choice", "0+rt": "launcher's choice, run-time tile size"}.get(w, w + " waves per workgroup"), "B": B,
                          "n": env.num_agents, "tile": env.tile_size, "obs_bytes": nb, "raster_ms": r,
                          "raster_frac_of_8TBps": nb / r / 1e6 / 8000, "step_ms": s, "step_frac_of_8TBps": nb / s / 1e6 / 8000,
                          "agent_steps_per_s": B * env.num_agents / s * 1e3}), flush=True)
    del env
    torch.cuda.empty_cache()
