#!/usr/bin/env python
"""Random-policy rollout on a batch of envs (needs an MI355X).

    python examples/random_rollout.py --batch 4096 --steps 200
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.envs import make, registered_envs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="MarlGrid-3AgentCluttered15x15-v0", choices=registered_envs)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()

env = make(args.env, batch_size=args.batch, auto_reset=True, strict=False)
obs = env.reset()                                             # (B, n, P, P, 3) uint8 on the GPU
n = env.num_agents
returns = torch.zeros(args.batch, n, device=obs.device)
episodes = torch.zeros((), dtype=torch.int64, device=obs.device)    # counted on the device: no host sync per step
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(args.steps):
    actions = torch.randint(0, 3, (args.batch, n), device=obs.device)     # left / right / forward
    obs, rew, done, _ = env.step(actions)
    returns += rew
    episodes += done.sum()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.check_errors()
episodes = int(episodes)
print("%s: %d envs x %d steps in %.3f s = %.1f M agent-steps/s; %d episodes finished; mean return %.4f"
      % (args.env, args.batch, args.steps, dt, args.batch * n * args.steps / dt / 1e6, episodes,
         float(returns.sum() / max(episodes, 1) / n)))
