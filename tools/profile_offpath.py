#!/usr/bin/env python
"""Workload for rocprofv3 counter passes over one obs-kernel instantiation: builds the case, 4 x mg_render_obs and
4 x env.step.  usage: profile_offpath.py tile5|tile6|tile8|tile11|prestige3|human"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.agents import GridAgentInterface  # noqa: E402
from marlgrid_amd.envs import ClutteredGoalCycleEnv, ClutteredMultiGrid  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "tile5"
B = 32768
if case.startswith("tile"):
    ts = int(case[4:])
    env = ClutteredMultiGrid(agents=[GridAgentInterface(color=c, view_size=7, view_tile_size=ts) for c in ("red", "blue", "purple")],
                             grid_size=15, clutter_density=0.15, batch_size=B, strict=False, auto_reset=True, place_obs=False)
else:
    n, ts = (3, 8) if case == "prestige3" else (1, 11)
    env = ClutteredGoalCycleEnv(agents=[GridAgentInterface(color="prestige", view_size=7, view_tile_size=ts, view_offset=1) for _ in range(n)],
                                grid_size=13, clutter_density=0.15, n_bonus_tiles=3, max_steps=250, respawn=True, reward_decay=False,
                                initial_reward=True, penalty=-1.5, batch_size=B, strict=False, auto_reset=True, place_obs=False)
env.reset()
a = torch.randint(0, 7, (B, env.num_agents)).to(env.device)
for _ in range(4):
    env.gen_obs()
for _ in range(4):
    env.step(a)
torch.cuda.synchronize()
