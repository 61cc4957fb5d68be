#!/usr/bin/env python
"""Follow-up to placement_probe.py: is the fast/slow split of obs buffers a property of HOW the memory was
allocated?  Raw hipMalloc allocations of several sizes (the obs size as torch rounds it, and exact powers of
two — one buddy block of the VRAM manager when memory is plentiful), the raster timed into each."""
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
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


def cost(ptr, iters=3):
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
    return ms.value


def raw(size):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), size)
    if rc != 0:
        raise RuntimeError("hipMalloc(%d) -> %d" % (size, rc))
    return p.value


MiB = 1 << 20
print("obs bytes %d = %.3f MiB" % (nbytes, nbytes / MiB))
for c in [r["obs"] for r in env._ring]:
    print("torch ring             ptr %#016x  %.4f ms" % (c.data_ptr(), cost(c.data_ptr())))
held = []
for label, size, count in (("raw obs-size(2MiB up)", (nbytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB), 10),
                           ("raw 1 GiB", 1 << 30, 10), ("raw 2 GiB", 2 << 30, 6), ("raw 4 GiB", 4 << 30, 4),
                           ("raw obs-size again", (nbytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB), 6)):
    ptrs = [raw(size) for _ in range(count)]
    held += ptrs
    for p in ptrs:
        line = "%-22s ptr %#016x  %.4f ms" % (label, p, cost(p))
        if size >= nbytes * 2:      # also the last obs-sized window of a bigger block
            q = (p + size - nbytes) // (2 * MiB) * (2 * MiB)
            line += "   tail window %#016x  %.4f ms" % (q, cost(q))
        print(line)
# are the verdicts stable?  re-time the first few
for p in held[:4]:
    print("re-timed               ptr %#016x  %.4f ms" % (p, cost(p, 10)))
for p in held:
    hip.hipFree(p)
