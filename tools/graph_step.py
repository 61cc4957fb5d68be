#!/usr/bin/env python
"""env.step() captured into a HIP graph (torch.cuda.graph): launch-bound small batches replay the
step / reset / raster kernels with one graph launch per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

for B in (16, 256, 4096):
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, obs_buffers=1)
    env.reset()
    n = env.num_agents
    acts = torch.randint(0, 7, (64, B, n), device=env.device)
    static = acts[0].clone()
    for i in range(5):
        env.step(static)                       # warm: the first step re-traces the reset program
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        env.step(acts[i % 64])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 300
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(static)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        static.copy_(acts[i % 64])
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 300
    print("B=%5d: eager %.1f us/step (%.3g agent-steps/s), graph replay %.1f us/step (%.3g)"
          % (B, eager * 1e6, B * n / eager, graph * 1e6, B * n / graph))
    env.check_errors()
    del env, g
