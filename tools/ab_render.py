#!/usr/bin/env python
"""In-process interleaved A/B of obs-render variants (MG_RENDER_VARIANT is read at every launch).
usage: ab_render.py [variants...]   e.g. ab_render.py 0 1 3 4"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the measurement build (marlgrid_amd/csrc/build.sh ab): the product library has no switches to flip
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
N.use_library(os.path.join(ROOT, "marlgrid_amd", "csrc", os.environ.get("AB_LIB", "libmarlgrid_hip_ab.so")))   # the measurement build
from marlgrid_amd.envs import make  # noqa: E402

variants = sys.argv[1:] or ["0", "2", "3", "4", "6", "11"]      # "V" or "V:waves_per_workgroup"
B = int(os.environ.get("B", "32768"))
if os.environ.get("TILE"):          # the bench scenario's shape with another view_tile_size
    from marlgrid_amd.agents import GridAgentInterface
    from marlgrid_amd.envs import ClutteredMultiGrid
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=int(os.environ.get("VIEW", "7")), view_tile_size=int(os.environ["TILE"])) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
else:
    env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
for i in range(30):
    env.step(torch.randint(0, 7, (B, env.num_agents), generator=g).cuda())
torch.cuda.synchronize()
vs, ts, n = env.view_size, env.tile_size, env.num_agents
alg = B * n * (vs * ts * vs * ts * 3 + vs * vs + 8 * n)
res = {v: [] for v in variants}
ms = C.c_float(0)
for rep in range(7):
    for v in variants:
        f0 = v.split(":")[0]
        os.environ["MG_RENDER_RASTER"] = "1" if f0 == "R" else "0"   # "R": the assemble-and-stream raster at tile 8
        os.environ["MG_RENDER_DEPTH"] = f0[1:] if f0.startswith("D") else "0"   # "D<k>": every wave looks k envs ahead
        os.environ["MG_RENDER_FRONT"] = "1" if f0[0] in "FM" else "0"   # "F" / "M<k>": the two-kernel dense-front experiment
        os.environ["MG_FRONT_MODE"] = f0[1:] if f0.startswith("M") else "0"   # "M<k>": dense-front measurement modes
        os.environ["MG_RENDER_VARIANT"] = "0" if (f0 in ("R", "E", "F") or f0[0] in "DM") else f0
        os.environ["MG_RENDER_WPB"] = v.split(":")[1] if ":" in v else "0"
        os.environ["MG_RENDER_PER_CU"] = v.split(":")[2] if v.count(":") > 1 else "0"   # "V:wpb:workgroups per CU"
        N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 40,
                                            C.byref(ms), env._stream()))
        res[v].append(ms.value)
for v in variants:
    m = statistics.median(res[v])
    print("variant %s: median %.4f ms (min %.4f max %.4f)  %.0f GB/s  %.1f%% of 8 TB/s"
          % (v, m, min(res[v]), max(res[v]), alg / m / 1e6, alg / m / 1e6 / 80))
