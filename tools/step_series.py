#!/usr/bin/env python
"""The launch duration of every env.step() over a few episodes (HIP events around every launch, read at the end):
what the auto-reset inside the step launch costs as the batch's episodes run into max_steps / goals.  Per step: ms,
the number of envs that finished (done), the number of agents that are done.  usage: step_series.py [steps] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 330
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
env = make(os.environ.get("WL", "MarlGrid-3AgentCluttered15x15-v0"), batch_size=B, auto_reset=True, strict=False)
env.reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 7, (B, env.num_agents), generator=g).to(env.device) for _ in range(64)]
for i in range(int(os.environ.get("WARM", "0"))):
    env.step(acts[i % 64])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
dones = torch.zeros(T, dtype=torch.int32, device=env.device)
ev[0].record()
for t in range(T):
    _o, _r, d, _ = env.step(acts[t % 64])
    ev[t + 1].record()
    dones[t] = d.sum()
torch.cuda.synchronize()
ms = [ev[t].elapsed_time(ev[t + 1]) for t in range(T)]
dn = dones.cpu().tolist()
print("step:ms(envs done)  [B = %d, kept buffers %s]" % (B, getattr(env._groups[0], "placement_ms", {}).get("kept")))
print(" ".join("%d:%.3f(%d)" % (t, ms[t], dn[t]) for t in range(T)))
quiet = sorted(m for m, d in zip(ms, dn) if d == 0)
print("steps with no finished env: %d, median %.4f ms; with 1..1%% of the envs: median %.4f; with more: median %.4f" % (
    len(quiet), quiet[len(quiet) // 2] if quiet else 0,
    (lambda v: sorted(v)[len(v) // 2] if v else 0)([m for m, d in zip(ms, dn) if 0 < d <= B // 100]),
    (lambda v: sorted(v)[len(v) // 2] if v else 0)([m for m, d in zip(ms, dn) if d > B // 100])))
