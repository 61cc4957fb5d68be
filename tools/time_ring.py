#!/usr/bin/env python
"""env.step() wall time with 1 / 2 / 3 output buffer sets (same process, interleaved)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

B = 32768
envs = {k: make("MarlGrid-3AgentCluttered15x15-v0", batch_size=B, auto_reset=True, strict=False, obs_buffers=k) for k in (1, 2, 3)}
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
for e in envs.values():
    e.reset()
    for i in range(30):
        e.step(acts[i % 16])
torch.cuda.synchronize()
for rep in range(3):
    for k, e in envs.items():
        t0 = time.perf_counter()
        for i in range(300):
            e.step(acts[i % 16])
        torch.cuda.synchronize()
        print("obs_buffers=%d: %.4f ms/step" % (k, (time.perf_counter() - t0) / 300 * 1e3))
