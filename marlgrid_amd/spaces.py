"""Minimal gym-style spaces (the image has no `gym`): attribute holders + sample()."""
import numpy as np


class Space(object):
    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape or ()), np.dtype(dtype)

    def sample(self):
        if np.issubdtype(self.dtype, np.integer):
            return np.random.randint(self.low, int(self.high) + 1, size=self.shape).astype(self.dtype)
        return np.random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __iter__(self):
        return iter(self.spaces)


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}
