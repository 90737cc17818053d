"""Host-side space metadata (API of rlpyt/spaces/: ``sample``, ``null_value``, ``shape``,
``dtype``, and ``n`` for discrete boxes)."""
from collections import namedtuple

import numpy as np


class Space:
    def sample(self):
        raise NotImplementedError

    def null_value(self):
        raise NotImplementedError


class IntBox(Space):
    """Integers in [low, high) of a given shape (rlpyt/spaces/int_box.py)."""

    def __init__(self, low, high, shape=None, dtype="int64", null_value=None):
        assert np.isscalar(low) and np.isscalar(high)
        self.low, self.high = low, high
        self.shape = () if shape is None else tuple(shape)
        self.dtype = np.dtype(dtype)
        assert np.issubdtype(self.dtype, np.integer)
        self._null_value = 0 if null_value is None else null_value

    def sample(self):
        return np.random.randint(low=self.low, high=self.high, size=self.shape,
                                 dtype=self.dtype)

    def null_value(self):
        null = np.zeros(self.shape, dtype=self.dtype)
        if self._null_value:
            null[...] = self._null_value
        return null

    @property
    def bounds(self):
        return self.low, self.high

    @property
    def n(self):
        return self.high - self.low

    def __repr__(self):
        return f"IntBox({self.low}-{self.high - 1} shape={self.shape})"


class FloatBox(Space):
    """Floats in [low, high] (rlpyt/spaces/float_box.py)."""

    def __init__(self, low, high, shape=None, null_value=0., dtype="float32"):
        self.dtype = np.dtype(dtype)
        if shape is None:
            self.low = np.asarray(low, dtype=self.dtype)
            self.high = np.asarray(high, dtype=self.dtype)
            self.shape = self.low.shape
        else:
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
        self._null_value = null_value

    def sample(self):
        return np.asarray(np.random.uniform(self.low, self.high, self.shape), dtype=self.dtype)

    def null_value(self):
        return np.full(self.shape, self._null_value, dtype=self.dtype)


class Composite(Space):
    """A namedtuple of sub-spaces (rlpyt/spaces/composite.py)."""

    def __init__(self, spaces, NamedTupleCls):
        self._spaces = list(spaces)
        self._NamedTupleCls = NamedTupleCls

    def sample(self):
        return self._NamedTupleCls(*(s.sample() for s in self._spaces))

    def null_value(self):
        return self._NamedTupleCls(*(s.null_value() for s in self._spaces))

    @property
    def shape(self):
        return self._NamedTupleCls(*(s.shape for s in self._spaces))

    @property
    def spaces(self):
        return self._spaces


EnvSpaces = namedtuple("EnvSpaces", ["observation", "action"])
