#!/usr/bin/env python
"""How fast is the size-generic (byte-granular) raster?  Default GridAgentInterface tiles are 5 px."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402

for vs, ts in ((7, 5), (7, 8), (7, 6), (5, 11), (7, 16)):
    B = 32768
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=vs, view_tile_size=ts) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False)
    env.reset()
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 20, C.byref(ms), env._stream()))
    nbytes = env.obs.numel()
    print("view %d tile %d: %.3f ms  %.0f GB/s" % (vs, ts, ms.value, nbytes / ms.value / 1e6))
    del env
