#!/usr/bin/env python
"""mg_encode's piece size on the measurement build (libmarlgrid_hip_ab.so reads MG_ENCODE_PC at every launch): 1 024 / 2 048 /
4 096 / 8 192 cells per workgroup, interleaved, HIP events around 100 launches each; results compared between the sizes.
usage: [B=32768] [WL=...] ab_encode.py"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marlgrid_amd import _native as N  # noqa: E402
N.use_library(os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so"))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = int(os.environ.get("B", "32768"))
env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False, place_obs=False)
env.reset()
g = torch.Generator().manual_seed(0)
for i in range(10):
    env.step(torch.randint(0, 7, (B, env.num_agents), generator=g).cuda())
outs = {}
res = {}
sizes = [0, 1024, 2048, 4096, 8192, 16384, "empty"]      # "empty": the launcher's shape with a kernel that returns at once
for rep in range(7):
    for pc in sizes:
        os.environ["MG_ENCODE_EMPTY"] = "1" if pc == "empty" else "0"
        if pc and pc != "empty":
            os.environ["MG_ENCODE_PC"] = str(pc)
        else:
            os.environ.pop("MG_ENCODE_PC", None)
        out = outs.setdefault(pc, torch.empty((B, env.width, env.height, 3), dtype=torch.uint8, device=env.device))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, out.data_ptr(), env._stream()))
        a.record()
        for i in range(100):
            N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, out.data_ptr(), env._stream()))
        b.record()
        b.synchronize()
        res.setdefault(pc, []).append(a.elapsed_time(b) / 100 * 1e3)
for pc in sizes[1:-1]:
    assert torch.equal(outs[pc], outs[0]), pc
nbytes = B * (env.cells_stride + 8 * env.num_agents + 3 * env.width * env.height)
for pc in sizes:
    m = statistics.median(res[pc])
    print("piece %5s cells: launch interval median %.2f us (min %.2f)  %.0f GB/s = %.3f of 8 TB/s" % (pc or "auto", m, min(res[pc]), nbytes / m / 1e3, nbytes / m / 1e3 / 8000))
