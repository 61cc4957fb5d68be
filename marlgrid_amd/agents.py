"""Agent interface + the README's learner wrappers.

`GridAgentInterface` (kandouss/marlgrid `marlgrid/agents.py:9-295`) is, in the batched engine, the
*configuration* of one agent slot (view geometry, colour, observation format, action enum and gym
spaces).  Per-env agent state (pos, dir, carrying, done, active) does not live on this object: it
is the packed record array `MultiGridEnv.agent_state` in HBM, one row per env.

`LearningAgent` / `IndependentLearners` exist only in the reference's README (`README.md:21-63`);
they are implemented here to that documented contract, over a leading env-batch dimension.
"""
from contextlib import contextmanager
from enum import IntEnum

import numpy as np

from . import spaces
from .objects import GridAgent, COLORS

_AF_PLACED = 4      # include/marlgrid_hip.h MG_AF_PLACED


class GridAgentInterface(GridAgent):
    class actions(IntEnum):        # agents.py:10-17
        left = 0
        right = 1
        forward = 2
        pickup = 3
        drop = 4
        toggle = 5
        done = 6

    def __init__(self, view_size=7, view_tile_size=5, view_offset=0, observation_style="image",
                 observe_rewards=False, observe_position=False, observe_orientation=False,
                 restrict_actions=False, see_through_walls=False, hide_item_types=[],
                 prestige_beta=0.95, prestige_scale=2, allow_negative_prestige=False, spawn_delay=0,
                 **kwargs):
        super().__init__(**kwargs)
        if observation_style not in ("image", "rich"):
            raise ValueError("%s kwarg 'observation_style' must be one of 'image', 'rich'."
                             % self.__class__.__name__)
        self.view_size = int(view_size)
        self.view_tile_size = int(view_tile_size)
        self.view_offset = int(view_offset)
        self.observation_style = observation_style
        self.observe_rewards = observe_rewards
        self.observe_position = observe_position
        self.observe_orientation = observe_orientation
        self.hide_item_types = list(hide_item_types)
        self.see_through_walls = bool(see_through_walls)
        self.restrict_actions = restrict_actions
        self.prestige_beta = 0.95 if prestige_beta > 1 else prestige_beta
        self.prestige_scale = prestige_scale
        self.allow_negative_prestige = allow_negative_prestige
        self.spawn_delay = int(spawn_delay)
        self.init_kwargs = dict(kwargs)

        P = self.view_tile_size * self.view_size
        image_space = spaces.Box(low=0, high=255, shape=(P, P, 3), dtype="uint8")   # agents.py:58-63
        if observation_style == "image":
            self.observation_space = image_space
        else:
            d = {"pov": image_space}
            if observe_rewards:
                d["reward"] = spaces.Box(low=-np.inf, high=np.inf, shape=(), dtype=np.float32)
            if observe_position:
                d["position"] = spaces.Box(low=0, high=1, shape=(2,), dtype=np.float32)
            if observe_orientation:
                d["orientation"] = spaces.Discrete(n=4)
            self.observation_space = spaces.Dict(d)
        self.action_space = spaces.Discrete(3 if restrict_actions else len(self.actions))
        self.metadata = {"color": self.color, "view_size": self.view_size,
                         "view_tile_size": self.view_tile_size}

    def clone(self):
        return self.__class__(
            view_size=self.view_size, view_tile_size=self.view_tile_size, view_offset=self.view_offset,
            observation_style=self.observation_style, observe_rewards=self.observe_rewards,
            observe_position=self.observe_position, observe_orientation=self.observe_orientation,
            restrict_actions=self.restrict_actions, see_through_walls=self.see_through_walls,
            hide_item_types=self.hide_item_types, prestige_beta=self.prestige_beta,
            prestige_scale=self.prestige_scale, allow_negative_prestige=self.allow_negative_prestige,
            spawn_delay=self.spawn_delay, color=self.color, **self.init_kwargs)

    def get_view_pos(self):
        """the agent's own cell inside its view (agents.py:233-234)"""
        return (self.view_size // 2, self.view_size - 1 - self.view_offset)

    # ---- per-env state and view geometry, batched --------------------------------------------------
    # The reference keeps pos / dir / done / active / carrying / prestige on the agent object and
    # offers geometry helpers around them (agents.py:141-288).  Here they are read-only views of the
    # env's packed agent records: tensors with a leading env dimension on the env's device.  `None`
    # results of the reference (an agent that is not on the grid, a cell outside the view) are -1.
    def _bind(self, env, k):
        self._env, self._k = env, k

    def _bound(self):
        env = getattr(self, "_env", None)
        if env is None or env._dry:
            raise RuntimeError("this agent is not attached to a live MultiGridEnv")
        return env, self._k

    @property
    def pos(self):
        """(B, 2) int64 — (-1, -1) where the agent is not on the grid (`agent.pos is None`)"""
        env, k = self._bound()
        import torch
        placed = ((env.agent_flags[:, k] & _AF_PLACED) != 0).unsqueeze(-1)
        return torch.where(placed, env.agent_pos[:, k], torch.full_like(env.agent_pos[:, k], -1))

    @property
    def dir(self):
        env, k = self._bound()
        return env.agent_dir[:, k]

    @property
    def done(self):
        env, k = self._bound()
        return env.agent_done[:, k]

    @property
    def active(self):
        env, k = self._bound()
        return env.agent_active[:, k]

    @property
    def carrying(self):
        """(B,) int64 object ids into `env.obj_reg.objs` (0: carrying nothing)"""
        env, k = self._bound()
        return env.agent_carrying[:, k]

    @property
    def prestige(self):
        """(B,) float64 (agents.py:141-153); tracked on the device only when some agent's colour is 'prestige'"""
        env, k = self._bound()
        if env.prestige_t is None:
            raise AttributeError("prestige is only tracked when an agent's colour is 'prestige'")
        return env.prestige_t[:, k]

    @staticmethod
    def _dir_vec(d):
        """forward vector per dir (agents.py:176-183): [(1,0),(0,1),(-1,0),(0,-1)][dir]"""
        import torch
        table = torch.tensor([[1, 0], [0, 1], [-1, 0], [0, -1]], dtype=torch.int64, device=d.device)
        return table[d.long() % 4]

    @property
    def dir_vec(self):
        return self._dir_vec(self.dir)

    @property
    def right_vec(self):
        """agents.py:185-191: (-dy, dx)"""
        import torch
        v = self.dir_vec
        return torch.stack([-v[..., 1], v[..., 0]], dim=-1)

    @property
    def front_pos(self):
        """the cell right in front of the agent (agents.py:193-198)"""
        return self.pos + self.dir_vec

    @staticmethod
    def _view_exts(pos, d, view_size, view_offset):
        """agents.py:237-266 on (…, 2) positions and (…,) directions -> (…, 4) topX, topY, botX, botY"""
        import torch
        x, y, h = pos[..., 0], pos[..., 1], view_size // 2
        d = d.long() % 4
        top_x = torch.where(d == 0, x - view_offset, torch.where(d == 2, x - view_size + 1 + view_offset, x - h))
        top_y = torch.where(d == 1, y - view_offset, torch.where(d == 3, y - view_size + 1 + view_offset, y - h))
        return torch.stack([top_x, top_y, top_x + view_size, top_y + view_size], dim=-1)

    def get_view_exts(self):
        """(B, 4): the square of cells the agent's view covers; bottom extents excluded"""
        return self._view_exts(self.pos, self.dir, self.view_size, self.view_offset)

    @staticmethod
    def _view_coords(pos, d, i, j, view_size, view_offset):
        """agents.py:200-230: absolute (i, j) -> the agent's view coordinates (may lie outside the view)"""
        import torch
        fwd = GridAgentInterface._dir_vec(d)
        dx, dy = fwd[..., 0], fwd[..., 1]
        rx, ry = -dy, dx
        ax = pos[..., 0] - 2 * view_offset * dx
        ay = pos[..., 1] - 2 * view_offset * dy
        tx = ax + dx * (view_size - 1) - rx * (view_size // 2)
        ty = ay + dy * (view_size - 1) - ry * (view_size // 2)
        lx, ly = torch.as_tensor(i, device=d.device) - tx, torch.as_tensor(j, device=d.device) - ty
        return rx * lx + ry * ly, -(dx * lx + dy * ly)

    def get_view_coords(self, i, j):
        """(vx, vy), each (B,): i, j are ints or (B,) tensors of absolute grid coordinates"""
        return self._view_coords(self.pos, self.dir, i, j, self.view_size, self.view_offset)

    def relative_coords(self, x, y):
        """(B, 2): view coordinates of cell (x, y), or (-1, -1) where it is outside the view
        (`None` in the reference, agents.py:268-278)"""
        import torch
        vx, vy = self.get_view_coords(x, y)
        inside = (vx >= 0) & (vy >= 0) & (vx < self.view_size) & (vy < self.view_size)
        out = torch.stack([vx, vy], dim=-1)
        return torch.where(inside.unsqueeze(-1), out, torch.full_like(out, -1))

    def in_view(self, x, y):
        """(B,) bool (agents.py:280-285)"""
        return (self.relative_coords(x, y) >= 0).all(dim=-1)

    def sees(self, x, y):
        raise NotImplementedError      # as upstream (agents.py:287-288)

    # value identity is wrong for agents: two red agents are two agents
    __eq__ = object.__eq__
    __hash__ = object.__hash__


class LearningAgent(GridAgentInterface):
    """README.md:21-27 — subclass and implement action_step / save_step
    (start_episode / end_episode optional).  All tensors carry a leading env-batch dimension."""

    def action_step(self, obs):
        raise NotImplementedError

    def save_step(self, *transition_values):
        raise NotImplementedError


class IndependentLearners(object):
    """README.md:29-63: a collection of agents that act and learn independently.

        agents = IndependentLearners(TestRLAgent(), TestRLAgent(), TestRLAgent())
        env = ClutteredMultiGrid(agents, grid_size=15, n_clutter=10, batch_size=4096)
        obs = env.reset()                                   # (B, n, P, P, 3) uint8
        with agents.episode():
            actions = agents.action_step(obs)               # (B, n)
            next_obs, rew, done, _ = env.step(actions)
            agents.save_step(obs, actions, next_obs, rew, done)
    """

    def __init__(self, *agents):
        self.agents = list(agents)

    def __iter__(self):
        return iter(self.agents)

    def __len__(self):
        return len(self.agents)

    def __getitem__(self, i):
        return self.agents[i]

    def action_step(self, obs_array):
        import torch
        acts = [agent.action_step(obs_array[:, k]) for k, agent in enumerate(self.agents)]
        acts = [a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)) for a in acts]
        dev = obs_array.device if torch.is_tensor(obs_array) else acts[0].device
        return torch.stack([a.to(dev).reshape(-1).long() for a in acts], dim=1)

    def save_step(self, obs, act, next_obs, rew, done):
        for k, agent in enumerate(self.agents):
            agent.save_step(obs[:, k], act[:, k], next_obs[:, k], rew[:, k], done)

    def start_episode(self):
        for agent in self.agents:
            if hasattr(agent, "start_episode"):
                agent.start_episode()

    def end_episode(self):
        for agent in self.agents:
            if hasattr(agent, "end_episode"):
                agent.end_episode()

    @contextmanager
    def episode(self):
        self.start_episode()
        try:
            yield self
        finally:
            self.end_episode()
