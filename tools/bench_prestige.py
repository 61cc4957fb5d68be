#!/usr/bin/env python
"""Obs-raster throughput of the goal-cycle scenario with 'prestige'-coloured agents (per-env recoloured tiles)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredGoalCycleEnv  # noqa: E402

B = 32768
for colors, ts in ((("prestige",) * 3, 8), (("red", "blue", "purple"), 8), (("prestige",), 11), (("red",), 11)):
    env = ClutteredGoalCycleEnv(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts, view_offset=1) for c in colors],
                                grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True,
                                batch_size=B, strict=False, auto_reset=True)
    env.reset()
    g = torch.Generator().manual_seed(0)
    for i in range(20):
        env.step(torch.randint(0, 7, (B, len(colors)), generator=g).cuda())
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 20, C.byref(ms), env._stream()))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(50):
        env.step(torch.randint(0, 7, (B, len(colors)), generator=g).cuda())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("%s tile %d: raster %.3f ms  %.0f GB/s; env.step %.3f ms = %.3g agent-steps/s"
          % ("+".join(colors), ts, ms.value, env.obs.numel() / ms.value / 1e6, dt * 1e3, B * len(colors) / dt))
    del env
