"""Minimal stand-in for `gym` (<= 0.21 API surface that marlgrid touches). Test-only."""
import importlib

from . import spaces, utils, core, envs  # noqa: F401
from .core import Env, Wrapper  # noqa: F401
from .envs.registration import register, registry  # noqa: F401


def make(env_id, **kwargs):
    entry = registry[env_id]
    mod_name, attr = entry.split(":")
    mod = importlib.import_module(mod_name)
    return getattr(mod, attr)(**kwargs)
