#!/usr/bin/env python
"""Workgroup shape of the obs kernel by batch size: builds of the library that force 4 / 8 / 16 waves per workgroup
(`make -C marlgrid_amd/csrc exp EXP=2|6`; EXP=4, 8 waves, needs the <7, 8, 8> instantiation that round 5 measured and dropped) against the product's own choice, interleaved in one process on the
one-launch step (mg_step_render) and on the raster alone (mg_render_obs), for B in a list.
usage: [WL=MarlGrid-3AgentCluttered11x11-v0] [BS=1024,2048,...] wpb_sweep.py product.so exp2.so exp6.so"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

names = sys.argv[1:]
WL = os.environ.get("WL", "MarlGrid-3AgentCluttered11x11-v0")
BS = [int(b) for b in os.environ.get("BS", "1024,2048,4096,8192,16384,32768").split(",")]
vp, i32 = C.c_void_p, C.c_int32
libs = {}
for nm in names:
    L = C.CDLL(os.path.abspath(nm))
    L.mg_step_render.argtypes = [C.POINTER(N.Config), C.POINTER(N.State), vp, i32, vp, C.POINTER(N.GenProgram), vp, vp]
    L.mg_step_render.restype = i32
    L.mg_render_obs.argtypes = [C.POINTER(N.Config), C.POINTER(N.State), vp, vp, vp, vp, vp]
    L.mg_render_obs.restype = i32
    L.mg_build_info.restype = C.c_char_p
    libs[nm] = L
    print("%s: %s" % (nm, L.mg_build_info().decode()), flush=True)
g = torch.Generator().manual_seed(0)
for B in BS:
    env = make(WL, batch_size=B, auto_reset=True, strict=False)
    n = env.num_agents
    acts = [torch.randint(0, 7, (B, n), generator=g).cuda() for _ in range(16)]
    env.reset()
    env.step(acts[0])
    obs_bytes = env.obs.numel()

    def step(L, i):
        rc = L.mg_step_render(C.byref(env._cfg), C.byref(env._state), acts[i % 16].data_ptr(), 8, env.rewards.data_ptr(),
                              C.byref(env._reset_prog), env.obs.data_ptr(), env._stream())
        assert rc == 0, rc

    def raster(L, i):
        rc = L.mg_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), None, None, None, env._stream())
        assert rc == 0, rc

    row = {"workload": WL, "B": B, "obs_MB": obs_bytes / 1e6}
    for what, fn in (("step", step), ("raster", raster)):
        res = {nm: [] for nm in names}
        for rep in range(9):
            for nm in names:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(libs[nm], 0)
                a.record()
                for i in range(100):
                    fn(libs[nm], i)
                b.record()
                b.synchronize()
                res[nm].append(a.elapsed_time(b) / 100)
        row[what + "_ms"] = {os.path.basename(nm): round(statistics.median(res[nm]), 5) for nm in names}
    env.check_errors()
    print(json.dumps(row), flush=True)
    del env, acts
    torch.cuda.empty_cache()
