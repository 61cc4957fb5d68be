#!/usr/bin/env python
"""One process, one env: the step as tools/ab_fused.py launches it (mg_step_render straight through ctypes, one obs buffer)
against env.step() (the Python host path: argument checks, the two-buffer ring, the error-flag poll).  [TILE=5] [B=32768]"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402

B, ts = int(os.environ.get("B", "32768")), int(os.environ.get("TILE", "5"))
env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                         grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True)
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
env.reset()
env.step(acts[0])
L = N.lib()


def direct(i):
    N.check(L.mg_step_render(C.byref(env._cfg), C.byref(env._state), acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(),
                             C.byref(env._reset_prog), env.obs.data_ptr(), env._stream()))


def host(i):
    env.step(acts[i % 16])


res = {"direct": [], "env.step": []}
for rep in range(9):
    for name, fn in (("direct", direct), ("env.step", host)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(0)
        a.record()
        for i in range(100):
            fn(i)
        b.record()
        b.synchronize()
        res[name].append(a.elapsed_time(b) / 100)
for k, v in res.items():
    print("%-9s median %.4f ms (min %.4f max %.4f)" % (k, statistics.median(v), min(v), max(v)))
