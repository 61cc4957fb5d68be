#!/usr/bin/env python
"""What the constructed obs-buffer placement (MultiGridEnv._place_obs_buffers) finds in THIS process: kept buffers,
candidates drawn, every candidate's ms, why it stopped.  usage: placement_stats.py [B] [tile] [GiB]
GiB > 0: one allocation of that size is made and freed first — afterwards the driver hands out memory front to back from
the blocks it got back, which is how a GPU whose memory was never allocated before behaves (the state the search's
window scan is for)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
front = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if front > 0:
    import time
    t0 = time.perf_counter()
    big = torch.empty(int(front * (1 << 30)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    del big
    torch.cuda.empty_cache()
    print(json.dumps({"front_GiB": front, "alloc_s": t1 - t0, "free_s": time.perf_counter() - t1}), flush=True)
env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                         grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
def show(env, what):
    pm = dict(env._groups[0].placement_ms)
    pm["all"] = [round(x, 4) for x in pm["all"]]
    pm["kept"] = [round(x, 4) for x in pm["kept"]]
    print(json.dumps({"what": what, "B": B, "tile": ts, **pm}), flush=True)


show(env, "first env of the process (mg_obs_place, defaults)")
# a second env of the same size while the first is alive (a new search), then one after the first is gone (its released
# buffers are remembered by the library: no search)
env2 = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                          grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
show(env2, "second env, the first still alive")
del env, env2
import gc  # noqa: E402
gc.collect()
env3 = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                          grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
show(env3, "third env, after the first two were released")
