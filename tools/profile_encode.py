#!/usr/bin/env python
"""Workload for a rocprofv3 --kernel-trace pass over mg_encode (MultiGrid.encode for the batch, base.py:196-214): `--iters`
launches at the bench shard (32 768 envs of MarlGrid-3AgentCluttered15x15-v0) and at BASELINE configs[1] / [4]'s shapes;
the kernel's own duration (not the launch-to-launch interval tools/bench_cases.py times) is what the trace gives."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()

import torch  # noqa: E402
from marlgrid_amd import _native as N  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

for wl, B in (("MarlGrid-3AgentCluttered15x15-v0", 32768), ("MarlGrid-3AgentCluttered11x11-v0", 4096), ("MarlGrid-3AgentCluttered15x15-v0", 262144)):
    env = make(wl, batch_size=B, auto_reset=True, strict=False, place_obs=False)
    env.reset()
    g = torch.Generator().manual_seed(0)
    for i in range(10):
        env.step(torch.randint(0, 7, (B, env.num_agents), generator=g).cuda())
    out = torch.empty((B, env.width, env.height, 3), dtype=torch.uint8, device=env.device)
    for i in range(args.iters):
        N.check(env._lib.mg_encode(C.byref(env._cfg), C.byref(env._state), None, out.data_ptr(), env._stream()))
    torch.cuda.synchronize()
    print("encoded", wl, B, "algorithmic bytes per launch", B * (env.cells_stride + 8 * env.num_agents + 3 * env.width * env.height))
    del env, out
