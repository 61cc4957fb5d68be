#!/usr/bin/env python
"""Which HBM allocations does the raster's write pattern like?  Allocates obs-sized candidates (as
MultiGridEnv._place_obs_buffers does), times the raster into each and prints address vs ms — looking for
the rule behind the two clusters (~0.158 ms and ~0.20 ms per launch at the bench workload)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, place_obs=False)
env.reset()
nbytes = env.obs.numel()
ms = C.c_float(0)


def cost(ptr):
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), 3, C.byref(ms), env._stream()))
    return ms.value


cands = [r["obs"] for r in env._ring] + [torch.empty_like(env.obs) for _ in range(14)]
for c in cands:
    p = c.data_ptr()
    print("separate allocation  ptr %#016x  GiB-offset %8.3f  %.4f ms" % (p, (p % (1 << 40)) / 2**30, cost(p)))
# one big arena, windows at different offsets inside it
arena = torch.empty(nbytes * 8, dtype=torch.uint8, device=env.device)
a0 = arena.data_ptr()
for k in range(0, 8):
    for off in (0,):
        p = a0 + k * nbytes + off
        p = (p + (1 << 21) - 1) // (1 << 21) * (1 << 21) if k else p
        if p + nbytes <= a0 + arena.numel():
            print("arena %#016x window %d  ptr %#016x  %.4f ms" % (a0, k, p, cost(p)))
