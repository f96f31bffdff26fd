"""Minimal stand-ins for gym.spaces.Box / Dict (gym is not installed on the GPU box).
Duck-type compatible with the attributes the reference reads (`shape`, `dtype`, `low`, `high`,
`spaces`, `sample()`), cf. serl/data/replay_buffer.py:17-28, serl/agents/sac/sac_learner.py:56-58."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        if np.issubdtype(self.dtype, np.integer):
            return self._rng.integers(self.low, self.high.astype(np.int64) + 1, dtype=np.int64).astype(self.dtype)
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box{self.shape}[{self.dtype}]"


class Dict:
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or {}, **kw)

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def __repr__(self):
        return f"Dict({self.spaces})"


def is_box(space) -> bool:
    return hasattr(space, "shape") and hasattr(space, "dtype") and not hasattr(space, "spaces")


def is_dict(space) -> bool:
    return hasattr(space, "spaces")
