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

    # ------------------------------------------------------------------ device-resident sampling
    def sample_jax(self, batch_size: int, keys: Optional[Iterable[str]] = None):
        """Mirrors serl/data/dataset.py:106-129: `key, rng = split(rng)`, indices from
        jax.random.randint(key, (batch_size,), 0, len(self)) and a row gather of every leaf, all on
        the device the leaves live on (torch CUDA tensors, e.g. GPUReplayBuffer); returns
        `(indx.max(), batch)`.  As in the reference the rng starts at PRNGKey(seed or 42), `keys` is
        accepted and ignored (the jitted closure takes the whole dataset_dict) and the batch size of
        the first call is the one every later call uses."""
        import torch
        from .. import ops, prng
        if not hasattr(self, "rng"):
            self.rng = prng.PRNGKey((self._seed or 42) & 0xFFFFFFFF)
            self._sample_jax_batch = int(batch_size)
        n = self._sample_jax_batch
        key, self.rng = prng.split(self.rng)
        leaves = []

        def walk(tree):
            if isinstance(tree, dict):
                return {k: walk(v) for k, v in tree.items()}
            if not (isinstance(tree, torch.Tensor) and tree.is_cuda):
                raise TypeError("sample_jax needs device-resident leaves (use GPUReplayBuffer)")
            out = torch.empty((n, *tree.shape[1:]), dtype=tree.dtype, device=tree.device)
            leaves.append((tree, out))
            return out

        batch = walk(self.dataset_dict)
        dev = leaves[0][0].device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            indx = torch.empty(n + 1, dtype=torch.int32, device=dev)
            ops.prng_randint(key, n, 0, len(self), indx.data_ptr(), indx.data_ptr() + 4 * n, stream)
            for src, out in leaves:
                assert src.is_contiguous()
                row_bytes = src[0].numel() * src.element_size()
                ops.take_rows(src.data_ptr(), src.shape[0], row_bytes, indx.data_ptr(), out.data_ptr(), n, stream)
        return indx[n], freeze(batch)

    def split(self, ratio: float):
        """serl/data/dataset.py:131-143: shuffled index split into two Datasets."""
        assert 0 < ratio < 1
        index = np.arange(len(self), dtype=np.int32)
        self.np_random.shuffle(index)
        cut = int(self.dataset_len * ratio)
        return Dataset(take(self.dataset_dict, index[:cut])), Dataset(take(self.dataset_dict, index[cut:]))

    def _trajectory_boundaries_and_returns(self):
        """serl/data/dataset.py:145-162: episode [start, end) rows and returns from the `dones` flags
        (a trailing unfinished episode has a start but no end / return)."""
        n = len(self)
        dones = np.asarray(self.dataset_dict["dones"][:n]).astype(bool)
        ends = (np.flatnonzero(dones) + 1).tolist()
        starts = [0] + [e for e in ends if e < n]
        rewards = self.dataset_dict["rewards"]
        returns = []
        for s, e in zip(starts, ends):
            acc = 0 if s == 0 else 0.0
            for r in rewards[s:e]:           # sequential float32 accumulation, as the reference loop
                acc += r
            returns.append(acc)
        return starts, ends, returns

    def filter(self, take_top: Optional[float] = None, threshold: Optional[float] = None):
        """serl/data/dataset.py:164-188: keep the episodes whose return reaches `threshold` (or the
        top `take_top` percent)."""
        assert (take_top is None) != (threshold is None)
        starts, ends, returns = self._trajectory_boundaries_and_returns()
        if take_top is not None:
            threshold = np.percentile(returns, 100 - take_top)
        keep = np.zeros(len(self), dtype=bool)
        for s, e, r in zip(starts, ends, returns):
            if r >= threshold:
                keep[s:e] = True
        self.dataset_dict = take(self.dataset_dict, keep)
        self.dataset_len = _check_lengths(self.dataset_dict)

    def normalize_returns(self, scaling: float = 1000):
        """serl/data/dataset.py:190-195."""
        _, _, returns = self._trajectory_boundaries_and_returns()
        self.dataset_dict["rewards"] /= np.max(returns) - np.min(returns)
        self.dataset_dict["rewards"] *= scaling
