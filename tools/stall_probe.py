#!/usr/bin/env python
"""Where does the one ~50 ms hiccup of a bench run come from?  Host time of every env.step() call (no events, no
syncs), then the same with the garbage collector frozen; the calls that took over 2 ms are listed."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = 32768
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True)
env.reset()
g = torch.Generator().manual_seed(0)
pool = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(64)]
for i in range(5):
    env.step(pool[i])
torch.cuda.synchronize()
for mode in ("gc on", "gc frozen", "gc on again"):
    if mode == "gc frozen":
        gc.collect(); gc.freeze(); gc.disable()
    elif mode == "gc on again":
        gc.enable()
    ts = []
    t_all = time.perf_counter()
    for i in range(3000):
        t = time.perf_counter()
        env.step(pool[i % 64])
        ts.append(time.perf_counter() - t)
        if i % 500 == 499:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
    slow = [(i, round(x * 1e3, 2)) for i, x in enumerate(ts) if x > 2e-3]
    print("%-12s 3000 steps in %.1f ms (%.4f ms per step); host calls over 2 ms: %s; gc counts %s" % (mode, t_all * 1e3, t_all / 3, slow[:12], gc.get_count()), flush=True)
