"""Attribute holders only: marlgrid constructs spaces but never samples them on the hot path."""
import numpy as np


class Space(object):
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, shape, np.dtype(dtype)


class Discrete(Space):
    def __init__(self, n):
        self.n = n


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)
