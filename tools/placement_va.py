#!/usr/bin/env python
"""The class of an observation buffer belongs to its virtual range (profiles/r03/README.md section 2): ONE set of
physical handles (the bench workload's 925 MB) mapped behind N virtual ranges in turn, the real raster timed
into each; then the first ranges again (is a range's class reproducible?).  (Round 3 also ran MultiGridEnv's
then-default range search three times here: 48 ranges tried per buffer, one fast range found in ~250.)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vmm_buffer import VmmBuffer as _LibBuffer, ab_lib  # noqa: E402  (sets MARLGRID_HIP_LIB: measurement build)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

NR = int(os.environ.get("RANGES", "40"))
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
env.reset()
L, ms = ab_lib(), C.c_float(0)


def raster(ptr, iters=4):
    N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
    return ms.value


mem = _LibBuffer(L, env.obs.numel(), env.device, 2 << 20)
assert mem.ok
rows = [(mem.ptr, raster(mem.ptr))]
t0 = time.perf_counter()
for i in range(NR - 1):
    assert mem.rebase()
    rows.append((mem.ptr, raster(mem.ptr)))
dt = time.perf_counter() - t0
for i, (p, c) in enumerate(rows):
    print("range %2d  va %#014x  (va>>21)%%512 %3d  raster %.4f ms" % (i, p, (p >> 21) % 512, c))
print("%d remaps + timings in %.3f s (%.1f ms each)" % (NR - 1, dt, dt / (NR - 1) * 1e3))
for i in (0, 1, 2, 3, 4, 5, 6, 7):
    mem.select(i)
    print("range %2d again  va %#014x  raster %.4f ms (first time %.4f)" % (i, mem.ptr, raster(mem.ptr), rows[i][1]))
mem2 = _LibBuffer(L, env.obs.numel(), env.device, 2 << 20)     # other physical memory behind the ranges' neighbours
print("a second buffer, as built: va %#014x raster %.4f ms" % (mem2.ptr, raster(mem2.ptr)))
del mem, mem2, env
