"""Host-side dataset base (mirrors serl/data/dataset.py:56-104): seeded numpy Generator and
index sampling.  Index generation deliberately stays on numpy's PCG64 Generator -- the same call
the reference makes (gym.utils.seeding.np_random(seed) == Generator(PCG64(SeedSequence(seed)))) --
so sampled indices are bit-identical by construction."""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np

from ..frozen_dict import FrozenDict, freeze


def _check_lengths(tree, n=None):
    for v in tree.values():
        if isinstance(v, dict):
            n = n or _check_lengths(v, n)
        elif isinstance(v, np.ndarray):
            n = n or len(v)
            assert n == len(v), "Inconsistent item lengths in the dataset."
        else:
            raise TypeError("Unsupported type.")
    return n


def take(tree, indx):
    if isinstance(tree, np.ndarray):
        return tree[indx]
    if isinstance(tree, dict):
        return {k: take(v, indx) for k, v in tree.items()}
    raise TypeError("Unsupported type.")


class Dataset:
    def __init__(self, dataset_dict, seed: Optional[int] = None):
        self.dataset_dict = dataset_dict
        self.dataset_len = _check_lengths(dataset_dict)
        self._np_random = None
        self._seed = None
        if seed is not None:
            self.seed(seed)

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed: Optional[int] = None) -> list:
        ss = np.random.SeedSequence(seed)
        self._seed = ss.entropy
        self._np_random = np.random.Generator(np.random.PCG64(ss))
        return [self._seed]

    def __len__(self) -> int:
        return self.dataset_len

    def _draw(self, batch_size):
        return self.np_random.integers(len(self), size=batch_size)

    def sample(self, batch_size: int, keys: Optional[Iterable[str]] = None, indx=None) -> FrozenDict:
        if indx is None:
            indx = self._draw(batch_size)
        if keys is None:
            keys = self.dataset_dict.keys()
        return freeze({k: take(self.dataset_dict[k], indx) for k in keys})
