#!/usr/bin/env python
"""Workload for rocprofv3 counter passes that separate the env step from the raster inside the ONE kernel of the
bench workload: 30 warm steps, then 6 x mg_render_obs (raster only) and 6 x mg_step_render (step + raster) — the
same kernel name, told apart by dispatch order (tools/step_counters.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = 32768
env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, place_obs=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).to(env.device) for _ in range(8)]
for i in range(30):
    env.step(acts[i % 8])
torch.cuda.synchronize()
for _ in range(6):
    env.gen_obs()
torch.cuda.synchronize()
for i in range(6):
    env.step(acts[i])
torch.cuda.synchronize()
