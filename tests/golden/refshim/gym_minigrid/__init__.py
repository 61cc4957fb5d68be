"""Stand-in for gym-minigrid: only `rendering` is used by marlgrid. Test-only."""
from . import rendering  # noqa: F401
