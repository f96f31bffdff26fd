"""Ring replay buffer (mirrors serl/data/replay_buffer.py:17-116) whose iterator hands out
device-resident batches (the reference `jax.device_put`s inside get_iterator, :88-91)."""
from __future__ import annotations

import collections
import threading
from typing import Optional

import numpy as np

from ..frozen_dict import FrozenDict, freeze
from ..spaces import is_box, is_dict
from .dataset import Dataset


def _alloc(space, capacity):
    if is_box(space):
        return np.zeros((capacity, *space.shape), dtype=space.dtype)   # zeros, not empty: wrapped demo windows (quirk C.7) stay deterministic
    if is_dict(space):
        return {k: _alloc(v, capacity) for k, v in space.spaces.items()}
    raise TypeError()


def _store(dst, src, at):
    if isinstance(dst, np.ndarray):
        dst[at] = src
    elif isinstance(dst, dict):
        assert dst.keys() == src.keys()
        for k in dst:
            _store(dst[k], src[k], at)
    else:
        raise TypeError()


def to_device(tree, device):
    """Host batch -> device tensors (pinned staging, async copy on the current stream)."""
    import torch
    if isinstance(tree, (dict, FrozenDict)):
        return freeze({k: to_device(v, device) for k, v in tree.items()})
    if isinstance(tree, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(tree))
        if device.type == "cuda":
            t = t.pin_memory().to(device, non_blocking=True)
        return t
    return tree


class ReplayBuffer(Dataset):
    def __init__(self, observation_space, action_space, capacity: int, next_observation_space=None):
        if next_observation_space is None:
            next_observation_space = observation_space
        data = dict(
            observations=_alloc(observation_space, capacity),
            next_observations=_alloc(next_observation_space, capacity),
            actions=np.zeros((capacity, *action_space.shape), dtype=action_space.dtype),
            rewards=np.zeros((capacity,), dtype=np.float32),
            masks=np.zeros((capacity,), dtype=np.float32),
            dones=np.zeros((capacity,), dtype=bool),
        )
        super().__init__(data)
        self._size = 0
        self._capacity = capacity
        self._insert_index = 0

    def __len__(self) -> int:
        return self._size

    def insert(self, data_dict):
        _store(self.dataset_dict, data_dict, self._insert_index)
        self._insert_index = (self._insert_index + 1) % self._capacity
        self._size = min(self._size + 1, self._capacity)

    def get_iterator(self, queue_size: int = 2, sample_args: dict = {}, device=None):
        """Prefetching iterator of device batches: `queue_size` batches are always in flight."""
        import torch
        dev = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        queue = collections.deque()

        def enqueue(n):
            for _ in range(n):
                queue.append(self._device_sample(dev, **sample_args))

        enqueue(queue_size)
        while queue:
            yield queue.popleft()
            enqueue(1)

    def _device_sample(self, dev, **sample_args):
        return to_device(self.sample(**sample_args), dev)

    def download(self, from_idx: int, to_idx: int):
        indices = np.arange(from_idx, to_idx)
        return to_idx, self.sample(batch_size=len(indices), indx=indices)

    def get_download_iterator(self):
        last_idx = 0
        while True:
            if last_idx >= self._size:
                raise RuntimeError(f"last_idx {last_idx} >= self._size {self._size}")
            last_idx, batch = self.download(last_idx, self._size)
            yield batch


class GPUReplayBuffer(ReplayBuffer):
    """Device-resident ring (mirrors serl/data/replay_buffer.py:118-152): every leaf lives in HBM,
    `upload` appends a block of transitions under a lock, `get_iterator` yields
    `(indx_max, batch)` drawn on the device by `sample_jax` (threefry randint + row gather
    kernels) with no host copy.  Unlike the reference's `copy_arrays_recursively`, nested
    observation dicts are written to their own leaves (the reference indexes the top-level dict
    with the nested key and only works for flat Box observations)."""

    def __init__(self, observation_space, action_space, capacity: int, next_observation_space=None, device=None):
        import torch
        super().__init__(observation_space, action_space, capacity, next_observation_space)
        if device is None and not torch.cuda.is_available():
            raise RuntimeError("GPUReplayBuffer needs a GPU")
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

        def put(tree):
            if isinstance(tree, dict):
                return {k: put(v) for k, v in tree.items()}
            return torch.from_numpy(tree).to(self._device)

        self.dataset_dict = put(self.dataset_dict)
        self.lock = threading.Lock()

    def _copy_block(self, dst, src, lo, hi):
        import torch
        if isinstance(dst, dict):
            for k in src.keys():
                self._copy_block(dst[k], src[k], lo, hi)
        elif isinstance(src, np.ndarray) or isinstance(src, torch.Tensor):
            t = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src
            dst[lo:hi].copy_(t.to(dst.dtype))
        else:
            raise TypeError()

    def insert(self, data_dict):
        with self.lock:
            self._copy_block(self.dataset_dict, _as_block(data_dict), self._insert_index, self._insert_index + 1)
            self._insert_index = (self._insert_index + 1) % self._capacity
            self._size = min(self._size + 1, self._capacity)

    def upload(self, data_dict):
        """Append len(data_dict['observations']) rows at the insert index (:141-147); like the
        reference's `.at[start:end].set`, a block that would run past the capacity is an error."""
        with self.lock:
            n = _leading_len(data_dict["observations"])
            lo = self._insert_index
            if lo + n > self._capacity:
                raise ValueError(f"upload of {n} rows at {lo} runs past the capacity {self._capacity}")
            self._copy_block(self.dataset_dict, data_dict, lo, lo + n)
            self._insert_index = (lo + n) % self._capacity
            self._size = min(self._size + n, self._capacity)

    def sample(self, batch_size: int, keys=None, indx=None):
        raise NotImplementedError("GPUReplayBuffer samples on the device: use sample_jax / get_iterator")

    def get_iterator(self, queue_size: int = 2, sample_args: dict = {}):
        while True:
            with self.lock:
                out = self.sample_jax(**sample_args)
            yield out


def _leading_len(tree):
    while isinstance(tree, (dict, FrozenDict)):
        tree = next(iter(tree.values()))
    return len(tree)


def _as_block(tree):
    if isinstance(tree, (dict, FrozenDict)):
        return {k: _as_block(v) for k, v in tree.items()}
    return np.asarray(tree)[None]
