"""Import the real reference (`/root/reference`, read-only) through the test-only shims.

Only usable in the build container. Used by `make_golden.py` and by the live-parity CPU tests
(which skip when `/root/reference` is absent, e.g. on the GPU box).
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "marlgrid"))


def load():
    """Returns the imported `marlgrid` reference package (with .base/.agents/.objects/.envs)."""
    if not available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    import numpy as np
    sys.dont_write_bytecode = True
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import marlgrid  # noqa: F401
    import marlgrid.base  # noqa: F401
    import marlgrid.agents  # noqa: F401
    import marlgrid.objects  # noqa: F401
    import marlgrid.envs  # noqa: F401
    assert marlgrid.__file__.startswith(REFERENCE_ROOT)
    return marlgrid
