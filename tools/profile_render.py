#!/usr/bin/env python
"""Workload for rocprofv3 passes (kernel-trace and, separately, PMC counters).

Runs, in order: known-size calibration kernels (1 GiB fill, 1 GiB copy) so that WRITE_SIZE /
FETCH_SIZE can be calibrated on this box, then `--iters` obs-render launches and `--iters` full
env steps of the bench workload (32 768 envs of MarlGrid-3AgentCluttered15x15-v0).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--batch", type=int, default=32768)
ap.add_argument("--workload", default="MarlGrid-3AgentCluttered15x15-v0")
args = ap.parse_args()

import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

dev = torch.device("cuda", 0)
GiB = 1 << 30
a = torch.empty(GiB, dtype=torch.uint8, device=dev)
b = torch.empty(GiB, dtype=torch.uint8, device=dev)
for _ in range(3):
    a.fill_(7)            # calibration: 1 GiB written
for _ in range(3):
    b.copy_(a)            # calibration: 1 GiB read + 1 GiB written
torch.cuda.synchronize()
del a, b

env = make(args.workload, batch_size=args.batch, device=dev, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (args.batch, env.num_agents), generator=g).to(dev) for _ in range(8)]
for i in range(args.iters):
    env.gen_obs()
for i in range(args.iters):
    env.step(acts[i % 8])
torch.cuda.synchronize()
print("profiled", args.workload, "batch", args.batch, "obs bytes", env.obs.numel())
