from ..base import MultiGridEnv, MultiGrid
from ..objects import Goal


class EmptyMultiGrid(MultiGridEnv):
    """Walled room, goal in the bottom-right corner (marlgrid/envs/empty.py)."""
    mission = "get to the green square"
    metadata = {}

    def _gen_grid(self, width, height):
        self.grid = MultiGrid((width, height))
        self.grid.wall_rect(0, 0, width, height)
        self.put_obj(Goal(color="green", reward=1), width - 2, height - 2)
