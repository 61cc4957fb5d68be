#!/usr/bin/env python
"""The reference README's training-loop skeleton (README.md:29-63), batched (needs an MI355X)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlgrid_amd.agents import IndependentLearners, LearningAgent  # noqa: E402
from marlgrid_amd.envs import ClutteredMultiGrid  # noqa: E402


class TestRLAgent(LearningAgent):
    """Acts at random and counts what it was asked to remember."""

    def __init__(self, **kw):
        super().__init__(view_tile_size=8, **kw)
        self.transitions = 0

    def action_step(self, obs):                      # obs: (B, P, P, 3) uint8
        return torch.randint(0, 3, (obs.shape[0],), device=obs.device)

    def save_step(self, obs, act, next_obs, rew, done):
        self.transitions += obs.shape[0]

    def start_episode(self):
        self.transitions = 0

    def end_episode(self):
        print("  %s agent saw %d transitions" % (self.color, self.transitions))


agents = IndependentLearners(TestRLAgent(color="red"), TestRLAgent(color="blue"), TestRLAgent(color="purple"))
env = ClutteredMultiGrid(agents, grid_size=15, n_clutter=10, batch_size=1024)

for i_episode in range(2):
    obs_array = env.reset()
    with agents.episode():
        episode_over = False
        while not episode_over:
            action_array = agents.action_step(obs_array)
            next_obs_array, reward_array, done, _ = env.step(action_array)
            agents.save_step(obs_array, action_array, next_obs_array, reward_array, done)
            obs_array = next_obs_array
            episode_over = bool(done.all())

# The same loop as a double-buffered sampler: the batch as TWO envs on two streams (MultiGridEnv.pipelined /
# make(id, pipeline=2) -> marlgrid_amd.sharding.ShardPipeline), stepped in turn.  While part 1's step kernel runs,
# part 0's policy and step are already queued on the other stream: the launches overlap (+10 % agent-steps/s at
# 32 768 envs on one MI355X).  Env g of the batch keeps the seed it has in the one big env.
pipe = ClutteredMultiGrid.pipelined(agents, parts=2, grid_size=15, n_clutter=10, batch_size=1024)
for i_episode in range(2):
    obs = pipe.reset()                                    # a list: part k's observations, ordered on pipe.streams[k]
    with agents.episode():
        over = [False] * pipe.parts
        while not all(over):
            for k in range(pipe.parts):
                with pipe.on(k):                          # part k's stream: its policy, its step, its bookkeeping
                    act = agents.action_step(obs[k])
                    nxt, rew, done, _ = pipe.step_part(k, act)
                    agents.save_step(obs[k], act, nxt, rew, done)
                    obs[k] = nxt
                    over[k] = bool(done.all())            # (a host sync per part and step: fine for an example)
pipe.check_errors()
