#!/usr/bin/env python
"""How often is an observation buffer of the bench workload in the fast class, by how it was obtained — all on
ONE box, interleaved: a fresh virtual range for the same physical memory (mg_obs_rebase), fresh library buffers
(2 MiB handles), raw hipMalloc, torch allocations.  N of each; raster ms per launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vmm_buffer import VmmBuffer as _LibBuffer, ab_lib  # noqa: E402  (sets MARLGRID_HIP_LIB: measurement build)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

NR = int(os.environ.get("N", "24"))
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
env.reset()
L, ms, nbytes, dev = ab_lib(), C.c_float(0), env.obs.numel(), env.device


def raster(ptr, iters=4):
    N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(ptr), iters, C.byref(ms), env._stream()))
    return ms.value


base = _LibBuffer(L, nbytes, dev, 2 << 20)
res = {"rebase": [raster(base.ptr)], "vmm-2M": [], "hipmalloc": [], "torch": []}
keep = []
for i in range(NR):
    assert base.rebase()
    res["rebase"].append(raster(base.ptr))
    m = _LibBuffer(L, nbytes, dev, 2 << 20)
    res["vmm-2M"].append(raster(m.ptr))
    h = _LibBuffer(L, nbytes, dev, 0)
    res["hipmalloc"].append(raster(h.ptr))
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    res["torch"].append(raster(t.data_ptr()))
    keep.append((m, h, t))          # everything stays allocated: every candidate is new memory / a new range
    if len(keep) > 40:
        keep.pop(0)
for k, v in res.items():
    s = sorted(v)
    print("%-10s n %2d  min %.4f  p25 %.4f  median %.4f  max %.4f  fast (< 0.172 ms): %d  | %s" % (
        k, len(v), s[0], s[len(s) // 4], s[len(s) // 2], s[-1], sum(x < 0.172 for x in v), " ".join("%.3f" % x for x in v)))
