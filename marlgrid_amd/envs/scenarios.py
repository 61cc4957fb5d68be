"""Scenario classes of the batched engine.

Each class describes a layout through `_gen_grid`, written against the same `MultiGrid` /
`put_obj` / `place_obj` vocabulary as upstream's `marlgrid/envs/{empty,cluttered,goalcycle,
viz_test}.py`.  Here `_gen_grid` does not build a grid: it is *recorded* (static cells -> a template,
`place_obj` calls -> an ordered rejection-sampling program) and the HIP reset kernel replays the
recording for every env of the batch with that env's own RNG (see `MultiGridEnv._trace_gen_grid`).

Behaviour kept from upstream on purpose: the base constructor resets once before a subclass has
stored its own parameters, so layout parameters are read with `getattr(self, name, default)` and
the constructor-time grid differs from the grid of the first user `reset()` (cluttered.py:13-33,
goalcycle.py:13-40) — this consumes RNG draws and is part of seed-for-seed parity.
"""
from ..base import MultiGrid, MultiGridEnv
from ..objects import BonusTile, Goal, Wall


def _clutter_count(env, n_clutter, clutter_density):
    """n_clutter xor clutter_density -> number of scattered wall blocks (cluttered.py:15-18)"""
    if clutter_density is not None:
        return int(clutter_density * (env.width - 2) * (env.height - 2))
    return n_clutter


def _require_one_of(n_clutter, clutter_density):
    if (n_clutter is None) == (clutter_density is None):
        raise ValueError("Must provide n_clutter xor clutter_density in environment config.")


class _WalledRoom(MultiGridEnv):
    """helpers shared by the shipped scenarios"""
    metadata = {}

    def _room(self, width, height):
        self.grid = MultiGrid((width, height))
        self.grid.wall_rect(0, 0, width, height)

    def _scatter(self, factory, count, max_tries=100):
        for _ in range(count):
            self.place_obj(factory(), max_tries=max_tries)

    def _corner_goal(self, width, height):
        self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)

    def _spawn_anywhere(self):
        """Every shipped upstream scenario ends `_gen_grid` with `self.agent_spawn_kwargs = {}`
        (empty.py:15, cluttered.py:35, goalcycle.py:50, viz_test.py:14): constructor-supplied spawn
        kwargs only survive in scenario classes that do not do this."""
        self.agent_spawn_kwargs = {}


class EmptyMultiGrid(_WalledRoom):
    """A walled room with the goal in the bottom-right corner."""
    mission = "get to the green square"

    def _gen_grid(self, width, height):
        self._room(width, height)
        self._corner_goal(width, height)
        self._spawn_anywhere()


class ClutteredMultiGrid(_WalledRoom):
    """A walled room with `n_clutter` (or `clutter_density` x interior cells) wall blocks scattered
    at random; the goal sits in the corner unless `randomize_goal`."""
    mission = "get to the green square"

    def __init__(self, *args, n_clutter=None, clutter_density=None, randomize_goal=False, **kwargs):
        _require_one_of(n_clutter, clutter_density)
        super().__init__(*args, **kwargs)          # resets once with the defaults read below
        self.n_clutter = _clutter_count(self, n_clutter, clutter_density)
        self.randomize_goal = randomize_goal

    def _gen_grid(self, width, height):
        self._room(width, height)
        if getattr(self, "randomize_goal", True):
            self.place_obj(Goal(color="green", reward=1), max_tries=100)
        else:
            self._corner_goal(width, height)
        self._scatter(Wall, getattr(self, "n_clutter", 0))
        self._spawn_anywhere()


class ClutteredGoalCycleEnv(_WalledRoom):
    """`n_bonus_tiles` yellow bonus tiles to be visited in cyclic order (objects.BonusTile), plus
    clutter.  reward_decay defaults to False here, as upstream."""
    mission = "Cycle between yellow goal tiles."

    def __init__(self, *args, reward=1, penalty=0.0, n_clutter=None, clutter_density=None, n_bonus_tiles=3,
                 initial_reward=True, cycle_reset=False, reset_on_mistake=False, reward_decay=False, **kwargs):
        _require_one_of(n_clutter, clutter_density)
        kwargs["reward_decay"] = reward_decay
        super().__init__(*args, **kwargs)
        self.n_clutter = _clutter_count(self, n_clutter, clutter_density)
        self.reward, self.penalty = reward, penalty
        self.initial_reward, self.reset_on_mistake = initial_reward, reset_on_mistake
        self.n_bonus_tiles = n_bonus_tiles
        self.bonus_tiles = []

    def _bonus_tile(self, bonus_id):
        return BonusTile(color="yellow", reward=self.reward, penalty=self.penalty, bonus_id=bonus_id,
                         n_bonus=self.n_bonus_tiles, initial_reward=self.initial_reward,
                         reset_on_mistake=self.reset_on_mistake)

    def _gen_grid(self, width, height):
        self._room(width, height)
        for bonus_id in range(getattr(self, "n_bonus_tiles", 0)):
            self.place_obj(self._bonus_tile(bonus_id), max_tries=100)
        self._scatter(Wall, getattr(self, "n_clutter", 0))
        self._spawn_anywhere()


class VisibilityTestEnv(_WalledRoom):
    """A room split by a long horizontal wall — for looking at occlusion."""
    mission = ""

    def _gen_grid(self, width, height):
        self._room(width, height)
        self.grid.horz_wall(0, height // 2, width - 3, obj_type=Wall)
        self._spawn_anywhere()
