#!/usr/bin/env python
"""mg_obs_place on a configuration whose raster is not bound by HBM writes (views 13 at 5-pixel tiles: ~3 TB/s): it must stop
after its baseline ("not HBM-bound") instead of searching for a class that is not there; the bench shard beside it."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, product_envs, warnings
warnings.simplefilter("error")
for name, B in (("Edge-3AgentCluttered15x15-view13-tile5", 8192), ("MarlGrid-3AgentCluttered15x15-v0", 32768)):
    t0 = time.perf_counter()
    env = product_envs.build(name, batch_size=B)
    pm = env.obs_placement[0]
    print(name, B, "stopped:", pm["stopped"], "found", pm["found"], "cands", pm["candidates"], "sec %.3f" % pm["seconds"], "ctor %.2f s" % (time.perf_counter() - t0), "GB/s %.0f" % (pm["buffer_bytes"] / min(pm["all"][:2]) / 1e6 if pm["all"] else -1))
    del env
