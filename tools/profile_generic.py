#!/usr/bin/env python
"""Workload for rocprofv3 counter passes over the size-generic obs raster (view 7, tile 5 by default)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402

ts = int(sys.argv[1]) if len(sys.argv) > 1 else 5
env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                         grid_size=15, clutter_density=0.15, batch_size=32768, strict=False)
env.reset()
for _ in range(4):
    env.gen_obs()
torch.cuda.synchronize()
