"""Stand-in for numba: `njit` that reproduces the *unchecked-indexing* behaviour the reference
relies on (`marlgrid/agents.py:304-306` reads one column past the mask when view_offset == 0).

Canonical semantics: an out-of-range read yields False and an out-of-range
write is dropped.  Implemented by re-binding the jitted function's globals so that arrays it
creates / receives are a lenient ndarray subclass.  Test-only.
"""
import types

import numpy as np

boolean = np.bool_


class _Lenient(np.ndarray):
    def _oob(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if len(idx) != self.ndim:
            return False
        for k, n in zip(idx, self.shape):
            if not isinstance(k, (int, np.integer)):
                return False
            if k < 0 or k >= n:
                return True
        return False

    def __getitem__(self, idx):
        if self._oob(idx):
            return False
        return np.ndarray.__getitem__(self, idx)

    def __setitem__(self, idx, val):
        if self._oob(idx):
            return
        np.ndarray.__setitem__(self, idx, val)


class _NpProxy(object):
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def zeros(*a, **k):
        return np.zeros(*a, **k).view(_Lenient)


def njit(fn=None, **_kw):
    def wrap(f):
        g = dict(f.__globals__)
        g["np"] = _NpProxy()
        inner = types.FunctionType(f.__code__, g, f.__name__, f.__defaults__, f.__closure__)

        def call(*args):
            args = [a.view(_Lenient) if isinstance(a, np.ndarray) else a for a in args]
            out = inner(*args)
            return np.asarray(out).view(np.ndarray) if isinstance(out, np.ndarray) else out
        call.__wrapped__ = f
        return call
    return wrap(fn) if callable(fn) else wrap
