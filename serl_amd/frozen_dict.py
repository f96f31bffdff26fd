"""Small immutable mapping with the flax.core.FrozenDict methods the reference uses on batches and
parameter trees (`copy(add_or_replace=)`, `pop`, `unfreeze`, `keys/items`), cf.
serl/utils/commons.py:21-27,46 and serl/data/dataset.py:104."""
from __future__ import annotations

from collections.abc import Mapping


class FrozenDict(Mapping):
    __slots__ = ("_d",)

    def __init__(self, *args, **kw):
        d = dict(*args, **kw)
        object.__setattr__(self, "_d", {k: (FrozenDict(v) if isinstance(v, dict) else v) for k, v in d.items()})

    def __getitem__(self, k):
        return self._d[k]

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __contains__(self, k):
        return k in self._d

    def __setattr__(self, *a):
        raise TypeError("FrozenDict is immutable")

    def __repr__(self):
        return f"FrozenDict({self._d!r})"

    def copy(self, add_or_replace=None):
        d = dict(self._d)
        if add_or_replace:
            d.update(add_or_replace)
        return FrozenDict(d)

    def pop(self, key):
        """Returns (new FrozenDict without key, value) like flax."""
        d = dict(self._d)
        v = d.pop(key)
        return FrozenDict(d), v

    def unfreeze(self):
        return unfreeze(self)


def freeze(d) -> FrozenDict:
    return d if isinstance(d, FrozenDict) else FrozenDict(d)


def unfreeze(d):
    if isinstance(d, Mapping):
        return {k: unfreeze(v) for k, v in d.items()}
    return d
