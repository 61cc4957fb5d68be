#!/usr/bin/env python
"""Map of the whole HBM: obs-sized raw allocations until ~90 % of the device is taken, the raster timed into
each in allocation order (the VRAM manager hands out physical memory roughly in order, so the sequence is a
walk through physical address space).  Prints one line per allocation and a run-length summary."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
env.reset()
nbytes = (env.obs.numel() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
ms = C.c_float(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


def cost(ptr, iters=2):
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
    return ms.value


free, total = torch.cuda.mem_get_info()
count = int(free * float(os.environ.get("FRACTION", "0.9")) // nbytes)
print("free %.1f GiB of %.1f; %d allocations of %.3f GiB" % (free / 2**30, total / 2**30, count, nbytes / 2**30))
ptrs, costs = [], []
for i in range(count):
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), nbytes) != 0:
        print("hipMalloc failed at", i)
        break
    ptrs.append(p.value)
    costs.append(cost(p.value))
for i, (p, c) in enumerate(zip(ptrs, costs)):
    print("%3d  %7.2f GiB allocated before  va %#014x  %.4f ms  %s" % (i, i * nbytes / 2**30, p, c, "FAST" if c < 0.172 else ("mid" if c < 0.19 else "")))
runs, cur = [], None
for c in costs:
    k = "F" if c < 0.172 else ("m" if c < 0.19 else "s")
    if cur and cur[0] == k:
        cur[1] += 1
    else:
        cur = [k, 1]
        runs.append(cur)
print("runs:", " ".join("%s%d" % (k, n) for k, n in runs))
for p in ptrs:
    hip.hipFree(p)
