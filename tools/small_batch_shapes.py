#!/usr/bin/env python
"""Small batches: FEWER waves with MORE envs each?  At B = 4 096 the product launches 4 096 waves of one env (16 per CU,
four per SIMD): every wave steps its env on ONE lane, four such dependent chains share a SIMD.  The measurement build's
switches (MG_RENDER_WPB = waves per workgroup, MG_RENDER_PER_CU = workgroups per CU) give the other shapes — e.g. 4-wave
workgroups, one per CU: 1 024 waves of four envs, one wave per SIMD.  Whole step (mg_step_render) and raster, interleaved.
usage: [WL=...] [BS=1024,2048,4096,8192] small_batch_shapes.py"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
N.use_library(os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so"))      # the measurement build
from marlgrid_amd.envs import make  # noqa: E402

WL = os.environ.get("WL", "MarlGrid-3AgentCluttered11x11-v0")
BS = [int(b) for b in os.environ.get("BS", "1024,2048,4096,8192").split(",")]
SHAPES = [("16", "0"), ("4", "0"), ("4", "1"), ("4", "2"), ("8", "1"), ("8", "2"), ("16", "1")]     # (waves per workgroup, workgroups per CU; 0 = the launcher's)
g = torch.Generator().manual_seed(0)
L = N.lib()
for B in BS:
    env = make(WL, batch_size=B, auto_reset=True, strict=False, place_obs=False)
    n = env.num_agents
    acts = [torch.randint(0, 7, (B, n), generator=g).cuda() for _ in range(16)]
    env.reset()
    env.step(acts[0])

    def step(i):
        N.check(L.mg_step_render(C.byref(env._cfg), C.byref(env._state), acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(),
                                 C.byref(env._reset_prog), env.obs.data_ptr(), env._stream()))

    def raster(i):
        N.check(L.mg_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), None, None, None, env._stream()))

    row = {"workload": WL, "B": B}
    for what, fn in (("step", step), ("raster", raster)):
        res = {s: [] for s in SHAPES}
        for rep in range(7):
            for s in SHAPES:
                os.environ["MG_RENDER_WPB"], os.environ["MG_RENDER_PER_CU"] = s
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(0)
                a.record()
                for i in range(100):
                    fn(i)
                b.record()
                b.synchronize()
                res[s].append(a.elapsed_time(b) / 100)
        row[what + "_ms"] = {"wpb %s x %s per CU" % s: round(statistics.median(res[s]), 5) for s in SHAPES}
    env.check_errors()
    print(json.dumps(row), flush=True)
    del env, acts
