#!/usr/bin/env python
"""Interleaved A/B of two builds of libmarlgrid_hip.so in ONE process (the GPU's clocks drift by tens
of percent within a minute of load, so separate processes cannot be compared):
    ab_two_libs.py path/to/other.so [reps]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

other = C.CDLL(os.path.abspath(sys.argv[1]))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
B = int(os.environ.get("B", "32768"))
env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False, place_obs=("search" if os.environ.get("PLACE", "1") != "0" else False))
env.reset()
g = torch.Generator().manual_seed(0)
for i in range(30):
    env.step(torch.randint(0, 7, (B, env.num_agents), generator=g).cuda())
torch.cuda.synchronize()
libs = {"this build": env._lib, "other": other}
for L in libs.values():
    L.mg_time_render_obs.restype = C.c_int32
res = {k: [] for k in libs}
ms = C.c_float(0)
for rep in range(reps):
    for k, L in libs.items():
        rc = L.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), C.c_void_p(env.obs.data_ptr()), 40,
                                  C.byref(ms), env._stream())
        assert rc == 0, rc
        res[k].append(ms.value)
nbytes = env.obs.numel()
for k in libs:
    m = statistics.median(res[k])
    print("%-10s median %.4f ms (min %.4f max %.4f)  %.0f GB/s   first..last: %.4f .. %.4f"
          % (k, m, min(res[k]), max(res[k]), nbytes / m / 1e6, res[k][0], res[k][-1]))
