"""Frame-deduplicated pixel replay (mirrors serl/data/memory_efficient_replay_buffer.py:13-169).

Same storage contract as the reference -- each camera frame is stored once per row, a validity mask
marks rows with S in-episode predecessors, sampling redraws invalid rows one by one -- but built
for a 288 GB HBM part: with `device="cuda"` the frame rows live in HBM (2e5 x 2 cameras x 48 KB =
19.7 GB), `insert` uploads one 48 KB frame per camera, and `get_iterator` yields batches whose pixel
leaves are lazy `FrameWindows` (indices only), so the 2 GB-per-update host gather + H2D copy of the
reference disappears: DrQLearner.update gathers, unpacks and crops straight out of the frame store.
With `device=None` frames stay in host numpy arrays exactly like the reference.
"""
from __future__ import annotations

import copy
from typing import Iterable, Optional, Tuple

import numpy as np

from ..frozen_dict import freeze
from ..spaces import Box, Dict
from .dataset import take
from .device_batch import FrameWindows
from .replay_buffer import ReplayBuffer, to_device


class MemoryEfficientReplayBuffer(ReplayBuffer):
    def __init__(self, observation_space, action_space, capacity: int, pixel_keys: Tuple[str, ...] = ("pixels",),
                 device=None):
        self.pixel_keys = tuple(pixel_keys)
        spaces = dict(observation_space.spaces)
        self._num_stack = None
        self._frame_shape = {}
        for k in self.pixel_keys:
            sp = spaces[k]
            if self._num_stack is None:
                self._num_stack = sp.shape[-1]
            else:
                assert self._num_stack == sp.shape[-1]
            self._frame_shape[k] = tuple(sp.shape[:-1])
            spaces[k] = Box(low=np.asarray(sp.low)[..., 0], high=np.asarray(sp.high)[..., 0], dtype=sp.dtype)
        self._device = None
        if device is not None:
            import torch
            self._device = torch.device(device)
        host_spaces = spaces if self._device is None else {k: v for k, v in spaces.items() if k not in self.pixel_keys}
        next_space = Dict({k: v for k, v in spaces.items() if k not in self.pixel_keys})
        self._first = True
        self._is_correct_index = np.full(capacity, False, dtype=bool)
        super().__init__(Dict(host_spaces), action_space, capacity, next_observation_space=next_space)
        self._frames = None
        self._pending = []          # weak references to lazy window batches the learner has not gathered yet
        import threading
        self._lock = threading.RLock()      # insert (pin + overwrite) vs. update (read `packed` .. enqueue the gather)
        if self._device is not None:
            import torch
            self._frames = {k: torch.zeros((capacity, *self._frame_shape[k]), dtype=torch.uint8, device=self._device)
                            for k in self.pixel_keys}

    # ------------------------------------------------------------------ insert (:54-90)
    def _push_row(self, row, frames, valid):
        at = self._insert_index
        self._is_correct_index[at] = valid
        if self._frames is not None:
            import torch
            with self._lock:
                self._protect_pending(at)
                for k, f in frames.items():
                    self._frames[k][at].copy_(torch.from_numpy(np.ascontiguousarray(f)))
        else:
            row = dict(row)
            row["observations"] = dict(row["observations"], **frames)
        ReplayBuffer.insert(self, row)

    def _protect_pending(self, at):
        """Row `at` of the device frame store is about to be overwritten: batches sampled earlier whose windows hold
        it, and which the learner has not gathered yet, are gathered now (the reference's batches are snapshots:
        memory_efficient_replay_buffer.py:155-160 copies at sample time)."""
        live = []
        for ref in self._pending:
            fw = ref()
            if fw is None or fw.consumed or fw.packed is not None:
                continue
            if fw.touches(at):
                fw.pin()
            else:
                live.append(ref)
        self._pending = live

    def _prune_pending(self):
        """Drop batches that are gone, gathered or pinned (a learner that samples without inserting -- offline training,
        bench loops -- must not grow the list)."""
        with self._lock:
            self._pending = [r for r in self._pending
                             if (fw := r()) is not None and not fw.consumed and fw.packed is None]

    def _row_at(self, idx):
        row = take(self.dataset_dict, idx)
        if self._frames is not None:
            return row, {k: self._frames[k][idx].cpu().numpy() for k in self.pixel_keys}
        obs = dict(row["observations"])
        frames = {k: obs.pop(k) for k in self.pixel_keys}
        row["observations"] = obs
        return row, frames

    def insert(self, data_dict):
        S = self._num_stack
        if self._insert_index == 0 and self._capacity == len(self) and not self._first:
            for idx in range(len(self) - S, len(self)):       # ring wrapped mid-episode: re-seed the head
                row, frames = self._row_at(idx)
                self._push_row(row, frames, False)
        row = dict(data_dict)
        row["observations"] = dict(data_dict["observations"])
        row["next_observations"] = dict(data_dict["next_observations"])
        obs_px = {k: np.asarray(row["observations"].pop(k)) for k in self.pixel_keys}
        nxt_px = {k: np.asarray(row["next_observations"].pop(k)) for k in self.pixel_keys}
        if self._first:                                         # episode start: the S stacked frames, invalid
            for i in range(S):
                self._push_row(row, {k: v[..., i] for k, v in obs_px.items()}, False)
        self._first = bool(row["dones"])
        self._push_row(row, {k: v[..., -1] for k, v in nxt_px.items()}, True)
        for i in range(S):                                      # rows about to be overwritten lose validity
            self._is_correct_index[(self._insert_index + i) % len(self)] = False

    # ------------------------------------------------------------------ sample (:92-169)
    def _sample_indices(self, batch_size, demo_batch_size=0, demo_size=0):
        rng, n = self.np_random, len(self)
        indx = rng.integers(n, size=batch_size)
        # the reference walks i = 0..batch_size-1 and redraws invalid slots with scalar draws; only
        # invalid slots consume draws, so walking just those (ascending) yields the same stream
        for i in np.flatnonzero(~self._is_correct_index[indx]):
            while not self._is_correct_index[indx[i]]:
                indx[i] = rng.integers(n)
        if demo_size and demo_batch_size:
            demo_idx = rng.integers(demo_size, size=demo_batch_size)
            indx[-len(demo_idx):] = demo_idx                    # tail overwrite, no validity check
        return indx

    def _non_pixel(self, indx, keys):
        if keys is None:
            keys = list(self.dataset_dict.keys())
        else:
            keys = list(keys)
            assert "observations" in keys
        keys.remove("observations")
        batch = {k: take(self.dataset_dict[k], indx) for k in keys}
        batch["observations"] = {k: v[indx] for k, v in self.dataset_dict["observations"].items()
                                 if k not in self.pixel_keys}
        return batch, keys

    def sample(self, batch_size: int, keys: Optional[Iterable[str]] = None, indx=None,
               pack_obs_and_next_obs: bool = False, demo_batch_size: int = 0, demo_size: int = 0):
        if indx is not None:
            raise NotImplementedError()
        indx = self._sample_indices(batch_size, demo_batch_size, demo_size)
        batch, keys = self._non_pixel(indx, keys)
        S = self._num_stack
        for k in self.pixel_keys:
            if self._frames is not None:
                import torch
                win = FrameWindows(self._frames[k], self._capacity,
                                   torch.from_numpy(indx.astype(np.int64)).to(self._device), S).materialize().cpu().numpy()
            else:
                frames = self.dataset_dict["observations"][k]
                win = np.lib.stride_tricks.sliding_window_view(frames, S + 1, axis=0)[indx - S]
            if pack_obs_and_next_obs:
                batch["observations"][k] = win
            else:
                batch["observations"][k] = win[..., :-1]
                if "next_observations" in keys:
                    batch["next_observations"][k] = win[..., 1:]
        return freeze(batch)

    def _device_sample(self, dev, **sample_args):
        """Iterator path: non-pixel leaves as device tensors, pixel leaves as lazy FrameWindows."""
        if self._frames is None:
            return to_device(self.sample(**sample_args), dev)
        import torch
        args = dict(sample_args)
        batch_size = args.pop("batch_size")
        args.pop("pack_obs_and_next_obs", None)                 # lazy windows are packed by construction
        indx = self._sample_indices(batch_size, args.pop("demo_batch_size", 0), args.pop("demo_size", 0))
        batch, _ = self._non_pixel(indx, args.pop("keys", None))
        assert not args, f"unknown sample args {sorted(args)}"
        out = to_device(batch, dev).unfreeze()
        ix = torch.from_numpy(indx.astype(np.int64)).pin_memory().to(self._device, non_blocking=True)
        import weakref
        S = self._num_stack
        first = indx.astype(np.int64) - S
        first[first < 0] += self._capacity - S                  # the negative-index wrap of the window view (C.7)
        self._prune_pending()
        for k in self.pixel_keys:
            fw = FrameWindows(self._frames[k], self._capacity, ix, S, first_rows=first, lock=self._lock)
            with self._lock:
                self._pending.append(weakref.ref(fw))
            out["observations"][k] = fw
        return freeze(out)
