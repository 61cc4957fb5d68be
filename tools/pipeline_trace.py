#!/usr/bin/env python
"""Workload for `rocprofv3 --kernel-trace`: 300 steps of a ShardPipeline (two envs of 16 384 on two streams), then 300
steps of one env of 32 768 — tools/pipeline_trace.sh turns the trace into how much consecutive launches overlap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402
from marlgrid_amd.sharding import ShardPipeline  # noqa: E402

B, WL = 32768, "MarlGrid-3AgentCluttered15x15-v0"
pipe = ShardPipeline(lambda **kw: make(WL, auto_reset=True, strict=False, **kw), B, parts=2)
one = make(WL, batch_size=B, auto_reset=True, strict=False)
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, 3), generator=g).cuda() for _ in range(16)]
parts = [[pipe.part(k, a).contiguous() for k in range(2)] for a in acts]
pipe.reset(); one.reset()
torch.cuda.synchronize()
for i in range(300):
    pipe.step(parts[i % 16])
torch.cuda.synchronize()
for i in range(300):
    one.step(acts[i % 16])
torch.cuda.synchronize()
