"""marlgrid_amd — MI355X-native batched MarlGrid step engine.

A from-scratch implementation of the hot path of kandouss/marlgrid (`MultiGridEnv.step` ->
`gen_obs` raster, `MultiGrid.encode`, `reset`) for a batch of B independent envs whose state lives
in HBM, with every per-step operation a hand-written HIP kernel for gfx950.

    from marlgrid_amd.envs import make
    env = make("MarlGrid-3AgentCluttered15x15-v0", batch_size=4096)
    obs = env.reset()                                  # (B, 3, 56, 56, 3) uint8 on the GPU
    obs, rew, done, _ = env.step(actions)              # actions: (B, 3) ints
"""
from .agents import GridAgentInterface, IndependentLearners, LearningAgent  # noqa: F401
from .base import MultiGrid, MultiGridEnv, release_obs_cache  # noqa: F401

__version__ = "0.1.0"
