"""Flat device parameter storage.

Every network (actor, critic, target critic) is ONE contiguous float32 HBM buffer plus a static
layout {path -> (offset, shape)}; optimizer moments and gradients are buffers of the same layout, so
Adam + Polyak over a whole network is a single streaming kernel launch and the encoder hand-over
critic -> actor (serl/utils/commons.py:32-47) is one device-to-device copy of a prefix.

Leaf names are the flax auto-names of the reference modules so that
`.params` trees exchange 1:1 with reference checkpoints' structure.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ..frozen_dict import FrozenDict

ALIGN = 64  # floats (256 B): keeps every leaf 16-B aligned for float4 access


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class Layout:
    def __init__(self):
        self.entries: "OrderedDict[tuple, tuple]" = OrderedDict()
        self.size = 0

    def add(self, path, shape):
        path = tuple(path)
        assert path not in self.entries, path
        shape = tuple(int(s) for s in shape)
        self.entries[path] = (self.size, shape)
        self.size += _round_up(int(np.prod(shape)) if shape else 1)
        return self

    def offset(self, *path):
        return self.entries[tuple(path)][0]

    def shape(self, *path):
        return self.entries[tuple(path)][1]

    def prefix_size(self, top_key):
        """Size (floats) of the leading run of leaves whose path starts with `top_key`."""
        end = 0
        for path, (off, shape) in self.entries.items():
            if path[0] != top_key:
                break
            end = off + _round_up(int(np.prod(shape)) if shape else 1)
        return end

    # ---- host <-> flat
    def flatten(self, tree) -> np.ndarray:
        flat = np.zeros(self.size, dtype=np.float32)
        seen = 0
        for path, (off, shape) in self.entries.items():
            leaf = tree
            for k in path:
                leaf = leaf[k]
            a = np.asarray(leaf.detach().cpu() if hasattr(leaf, "detach") else leaf, dtype=np.float32)
            if a.shape != shape:
                raise ValueError(f"parameter {'/'.join(path)}: shape {a.shape} != {shape}")
            flat[off:off + a.size] = a.ravel()
            seen += 1
        if seen != _count_leaves(tree):
            raise ValueError("parameter tree has leaves the network does not define")
        return flat

    def unflatten(self, flat, wrap=lambda x: x):
        tree = {}
        for path, (off, shape) in self.entries.items():
            n = int(np.prod(shape)) if shape else 1
            d = tree
            for k in path[:-1]:
                d = d.setdefault(k, {})
            d[path[-1]] = wrap(flat[off:off + n].reshape(shape))
        return tree


def _count_leaves(tree):
    if hasattr(tree, "items"):
        return sum(_count_leaves(v) for _, v in tree.items())
    return 1


class ParamStore:
    """A flat device tensor + its layout."""

    def __init__(self, layout: Layout, flat):
        self.layout = layout
        self.flat = flat
        self.base = flat.data_ptr()

    def ptr(self, *path) -> int:
        return self.base + 4 * self.layout.entries[tuple(path)][0]

    def clone(self) -> "ParamStore":
        return ParamStore(self.layout, self.flat.clone())

    def zeros_like(self) -> "ParamStore":
        import torch
        return ParamStore(self.layout, torch.zeros_like(self.flat))

    def tree(self) -> FrozenDict:
        """FrozenDict of device views (no copies), flax-named."""
        return FrozenDict(self.layout.unflatten(self.flat))

    def numpy_tree(self) -> dict:
        host = self.flat.detach().cpu().numpy()
        return self.layout.unflatten(host, wrap=lambda a: a.copy())

    @classmethod
    def from_tree(cls, layout: Layout, tree, device) -> "ParamStore":
        import torch
        return cls(layout, torch.from_numpy(layout.flatten(tree)).to(device))
