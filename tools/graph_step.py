#!/usr/bin/env python
"""env.step() eager vs captured into a HIP graph (torch.cuda.graph) and replayed — VERDICT r03 item 5: is the small /
mid-batch step (BASELINE configs[1]: 4 096 envs of 3AgentCluttered11x11) host-bound, i.e. would a graph per ring slot
pay?  Per batch size: the raster alone (mg_time_render_obs), the eager step (GPU-paced loop), the host's own time per
eager step() call (how fast Python can issue: no sync), a graph replay with the actions copied into the graph's static
buffer (what a product graph-step would have to do), and a graph replay alone (the floor)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

WL = os.environ.get("WL", "MarlGrid-3AgentCluttered11x11-v0")
IT = 400
for B in (256, 1024, 4096, 16384, 32768):
    env = make(WL, batch_size=B, auto_reset=True, strict=False, obs_buffers=1)
    env.reset()
    n = env.num_agents
    acts = torch.randint(0, 7, (64, B, n), device=env.device)
    static = acts[0].clone()
    for i in range(5):
        env.step(static)                       # warm: the first step re-traces the reset program
    ms = C.c_float(0)
    N.check(env._lib.mg_time_render_obs(C.byref(env._cfg), C.byref(env._state), env.obs.data_ptr(), 50, C.byref(ms), env._stream()))
    raster = ms.value * 1e-3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(IT):
        env.step(acts[i % 64])
    host = (time.perf_counter() - t0) / IT       # the host's issue time (the queue absorbs it while it is not full)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / IT
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(static)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(IT):
        static.copy_(acts[i % 64])
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / IT
    t0 = time.perf_counter()
    for i in range(IT):
        g.replay()
    torch.cuda.synchronize()
    floor = (time.perf_counter() - t0) / IT
    print("%s B=%5d: raster alone %.1f us | eager step %.1f us (host issue %.1f us) | graph replay + actions copy %.1f us | "
          "graph replay alone %.1f us" % (WL, B, raster * 1e6, eager * 1e6, host * 1e6, graph * 1e6, floor * 1e6), flush=True)
    env.check_errors()
    del env, g
