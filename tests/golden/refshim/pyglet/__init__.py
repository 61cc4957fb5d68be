"""Empty stand-in: marlgrid/rendering.py imports pyglet unconditionally; never used headless."""
from . import gl, window  # noqa: F401
