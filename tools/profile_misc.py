#!/usr/bin/env python
"""Kernels that are built but are not the bench workload, one after the other, each timed with HIP events
(and meant to run under `rocprofv3 --kernel-trace --stats`, whose per-kernel averages must agree): the obs
raster on the other BASELINE configs and off the fast path (tile 5 / 6 / 11, run-time view size, atlas in
global memory, 'prestige' recolouring), mg_encode, mg_render_frame.  Prints one JSON line per case with the
algorithmic bytes per launch, the measured time and the fraction of the 8 TB/s HBM roofline."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredGoalCycleEnv, ClutteredMultiGrid, EmptyMultiGrid  # noqa: E402

COLS = ["red", "blue", "purple", "orange", "olive", "pink", "cyan", "yellow"]


def timed(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def raster_case(label, env, iters=20):
    env.reset()
    g = torch.Generator().manual_seed(0)
    for i in range(10):
        env.step(torch.randint(0, 7, (env.batch_size, env.num_agents), generator=g).to(env.device))
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), iters, C.byref(ms),
                                        env._stream()))
    acts = [torch.randint(0, 7, (env.batch_size, env.num_agents), generator=g).to(env.device) for _ in range(8)]
    step_ms = timed(lambda: env.step(acts[0]), iters)
    nb = env.obs.numel()
    print(json.dumps({"case": label, "kernel": "mg_render_obs", "B": env.batch_size, "n": env.num_agents,
                      "view": env.view_size, "tile": env.tile_size, "bytes_per_launch": nb, "ms": ms.value,
                      "GBps": nb / ms.value / 1e6, "frac_of_8TBps": nb / ms.value / 1e6 / 8000,
                      "env_step_ms": step_ms, "agent_steps_per_s": env.batch_size * env.num_agents / step_ms * 1e3}), flush=True)


def agents(n, vs, ts, **kw):
    return [GridAgentInterface(color=COLS[k % 8], view_size=vs, view_tile_size=ts, **kw) for k in range(n)]


cases = [
    ("3AgentCluttered11x11 B=4096 (BASELINE configs[1])", lambda: ClutteredMultiGrid(agents=agents(3, 7, 8), grid_size=11, clutter_density=0.15, batch_size=4096, strict=False, auto_reset=True)),
    ("4AgentEmpty9x9 B=65536 (configs[2])", lambda: EmptyMultiGrid(agents=agents(4, 7, 8), grid_size=9, batch_size=65536, strict=False, auto_reset=True)),
    ("8AgentCluttered30x30 view 9 B=131072 (configs[4] per GPU)", lambda: ClutteredMultiGrid(agents=agents(8, 9, 8), grid_size=30, clutter_density=0.15, batch_size=131072, strict=False, auto_reset=True)),
    ("tile 5 (GridAgentInterface default), view 7, B=32768", lambda: ClutteredMultiGrid(agents=agents(3, 7, 5), grid_size=15, clutter_density=0.15, batch_size=32768, strict=False, auto_reset=True)),
    ("tile 6, view 7", lambda: ClutteredMultiGrid(agents=agents(3, 7, 6), grid_size=15, clutter_density=0.15, batch_size=32768, strict=False, auto_reset=True)),
    ("tile 11, view 7", lambda: ClutteredMultiGrid(agents=agents(3, 7, 11), grid_size=15, clutter_density=0.15, batch_size=32768, strict=False, auto_reset=True)),
    ("tile 16, view 7", lambda: ClutteredMultiGrid(agents=agents(3, 7, 16), grid_size=15, clutter_density=0.15, batch_size=16384, strict=False, auto_reset=True)),
    ("tile 8, view 11 (run-time view size)", lambda: ClutteredMultiGrid(agents=agents(3, 11, 8), grid_size=15, clutter_density=0.15, batch_size=16384, strict=False, auto_reset=True)),
    ("tile 32, view 7 (atlas read from global memory)", lambda: ClutteredMultiGrid(agents=agents(3, 7, 32), grid_size=15, clutter_density=0.15, batch_size=4096, strict=False, auto_reset=True)),
    ("goal cycle, 3 'prestige' agents, tile 8", lambda: ClutteredGoalCycleEnv(agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=8, view_offset=1) for _ in range(3)], grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, batch_size=32768, strict=False, auto_reset=True)),
    ("goal cycle, 1 'prestige' agent, tile 11", lambda: ClutteredGoalCycleEnv(agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=11, view_offset=1)], grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, batch_size=32768, strict=False, auto_reset=True)),
]
only = sys.argv[1:]
for label, mk in cases:
    if only and not any(o in label for o in only):
        continue
    env = mk()
    raster_case(label, env)
    if label.startswith("3AgentCluttered11x11") or "configs[2]" in label:
        pass
    del env
    torch.cuda.empty_cache()

# mg_encode and mg_render_frame on the bench workload
env = ClutteredMultiGrid(agents=agents(3, 7, 8), grid_size=15, clutter_density=0.15, batch_size=32768, strict=False, auto_reset=True)
env.reset()
ms = timed(lambda: env.grid.encode(), 20)
nb = env.batch_size * env.width * env.height * 4       # 1 B read + 3 B written per cell
print(json.dumps({"case": "mg_encode, 32768 envs 15x15", "kernel": "mg_encode", "bytes_per_launch": nb, "ms": ms,
                  "GBps": nb / ms / 1e6, "frac_of_8TBps": nb / ms / 1e6 / 8000}), flush=True)
ids = list(range(64))
ms = timed(lambda: env.render(env_ids=ids, show_agent_views=False), 20)
nb = 64 * (env.height * 32) * (env.width * 32) * 3
print(json.dumps({"case": "mg_render_frame, 64 envs at 32 px tiles", "kernel": "mg_render_frame", "bytes_per_launch": nb,
                  "ms": ms, "GBps": nb / ms / 1e6, "frac_of_8TBps": nb / ms / 1e6 / 8000,
                  "note": "64 workgroups: a debugging / video path, far too small to fill the chip"}), flush=True)
