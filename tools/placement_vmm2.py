#!/usr/bin/env python
"""Is an observation buffer's write-rate class a property of its physical handles, or of how the raster's
streams fall onto them?  Measurement build.  N buffers of the bench workload built from 2 MiB handles are timed
with the real raster; then, for the slowest S and the fastest F:
  * S with its handles in reversed / randomly permuted order (same physical memory, same virtual range);
  * S and F after trading their even-numbered handles, then (again from the originals) their first halves;
  * buffers built from bigger handles (4 / 8 / 16 / 64 MiB).
One line each."""
import ctypes as C
import os
import sys

os.environ["MARLGRID_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
from vmm_buffer import VmmBuffer as _LibBuffer, ab_lib  # noqa: E402  (sets MARLGRID_HIP_LIB: measurement build)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=32768, auto_reset=True, strict=False, place_obs=False)
env.reset()
dev, nbytes, L = env.device, env.obs.numel(), ab_lib()
ms = C.c_float(0)
vp = C.c_void_p


def raster(mem, iters=6):
    N.check(L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), vp(mem.ptr), iters, C.byref(ms), env._stream()))
    return ms.value


def build(chunk):
    m = _LibBuffer(L, nbytes, dev, chunk)
    assert m.ok
    return m


NB = int(os.environ.get("NB", "12"))
bufs = [build(2 << 20) for _ in range(NB)]
cost = [raster(b) for b in bufs]
cost2 = [raster(b) for b in bufs]
for i, (c, c2) in enumerate(zip(cost, cost2)):
    print("vmm-2M buffer %2d  raster %.4f / %.4f ms" % (i, c, c2))
order = np.argsort(cost)
F, S = bufs[order[0]], bufs[order[-1]]
nh = F.info()["handles"]
print("fastest %d (%.4f), slowest %d (%.4f); %d handles each" % (order[0], cost[order[0]], order[-1], cost[order[-1]], nh))
rng = np.random.RandomState(0)


def permute(mem, perm, what):
    p = np.ascontiguousarray(perm, np.int32)
    N.check(L.mg_ab_obs_permute(vp(mem._h), vp(p.ctypes.data)))
    print("%-58s raster %.4f ms" % (what, raster(mem)), flush=True)
    inv = np.argsort(p).astype(np.int32)
    N.check(L.mg_ab_obs_permute(vp(mem._h), vp(inv.ctypes.data)))   # back to the original order


print("%-58s raster %.4f ms" % ("S as built", raster(S)))
permute(S, np.arange(nh)[::-1], "S, handles in reversed order")
for k in range(3):
    permute(S, rng.permutation(nh), "S, handles in random order %d" % k)
permute(S, np.roll(np.arange(nh), 1), "S, handles rotated by one (2 MiB)")
permute(S, np.roll(np.arange(nh), 7), "S, handles rotated by seven")
print("%-58s raster %.4f ms" % ("S, original order restored", raster(S)))
print("%-58s raster %.4f ms" % ("F as built", raster(F)))
permute(F, np.arange(nh)[::-1], "F, handles in reversed order")
permute(F, rng.permutation(nh), "F, handles in random order")


def exchange(slots, what):
    s = np.ascontiguousarray(slots, np.int32)
    N.check(L.mg_ab_obs_exchange(vp(S._h), vp(F._h), vp(s.ctypes.data), len(s)))
    print("%-58s S %.4f  F %.4f ms" % (what, raster(S), raster(F)), flush=True)
    N.check(L.mg_ab_obs_exchange(vp(S._h), vp(F._h), vp(s.ctypes.data), len(s)))   # and back


exchange(np.arange(0, nh, 2), "S and F trade their even-numbered handles")
exchange(np.arange(0, nh // 2), "S and F trade their first halves")
exchange(np.arange(0, nh // 4), "S and F trade their first quarters")
print("%-58s S %.4f  F %.4f ms" % ("originals restored", raster(S), raster(F)))
del bufs, F, S
for mib in (4, 8, 16, 64):
    ms_ = []
    keep = []
    for _ in range(6):
        m = build(mib << 20)
        keep.append(m)
        ms_.append(raster(m))
    print("vmm-%dM: %s" % (mib, " ".join("%.4f" % v for v in ms_)), flush=True)
    del keep
